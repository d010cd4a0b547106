#!/bin/bash
# Runs ON THE GPU BOX: the small-batch site kernels at config 5's two extreme shapes (2 x [28, 100352], 2 x [28, 802816]) for several
# builds of the library (tools/build_variant.sh; e.g. -DALIGNQ_S1_CAP=n: workgroups of the forward per batch slice).
cd "$GRAFT_REPO_ROOT"
for so in "" $@; do
  echo "== ${so:-product}"
  ALIGNQ_SO=$so python3 tools/office_shapes.py 2>/dev/null | grep "^bn_site" | python3 -c "
import sys, json
for l in sys.stdin:
    n, j = l.split(' ', 1); d = json.loads(j)
    print(n, 'site_fwd_kernel_us', d['site_fwd_kernel_us'], 'frac', d['site_fwd_kernel_frac_of_8TBs'], '| fwd_us', d['fwd_us'], '| site_bwd_kernel_us', d['site_bwd_kernel_us'])"
done
