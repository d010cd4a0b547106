for f in 1024 2048 3072 4096; do for b in 768 1536 2048 2304 3072; do
  echo -n "F=$f B=$b: "; ALIGNQ_S1_GRID_F=$f ALIGNQ_S1_GRID_B=$b python tools/roofline_shapes.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(round(v['fwd_us'],1),round(v['bwd_us'],1)) for k,v in d.items() if k.startswith('site_28')})"
done; done
