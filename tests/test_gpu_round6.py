"""Round 6: the benchmarked HIP-graph replay is the SAME computation as the oracle-checked eager step.

Every kernel of this repository reduces in a fixed order (per-workgroup partials, last-arriver tickets that fix
the summation order, no float atomics), so from one initial state `N` eager iterations and `capture(warmup=w)` + `N - w` replays must
leave identical bits in every parameter, momentum buffer, batch-norm running statistic, ADMM.D, alterD and gamma.  These tests
assert exactly that on the configurations bench.py times (reference iterations: cdf_alignment_admm/resnet-20-cifar-10/main.py:
288-378 and cdf_alignment_admm/dann_office/main.py:343-456), and print the first differing tensor when it does not hold."""
import numpy as np
import pytest
import torch

from tests.golden.det_init import det_init_

pytestmark = pytest.mark.gpu


def npy(t):
    return t.detach().float().cpu().numpy()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def full_state(model, step, admms):
    """Everything an iteration leaves behind, by name."""
    st = {}
    for n_, p in model.named_parameters():
        st["param:" + n_] = npy(p)
    for n_, b in model.named_buffers():
        st["buffer:" + n_] = npy(b)
    names = {id(p): n_ for n_, p in model.named_parameters()}
    for p, s in step.optimizer_t.state.items():
        if "momentum_buffer" in s and s["momentum_buffer"] is not None:
            st["momentum:" + names[id(p)]] = npy(s["momentum_buffer"])
    for i, a in enumerate(admms):
        if a.D is not None:
            st["D:%d" % i] = npy(a.D)
    return st


def same_bits(a, b):
    """Equal as fp32 values (+0 == -0: a zero-filled gradient accumulates -0 to +0) with NaNs in the same places."""
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def differing(sa, sb):
    assert set(sa) == set(sb), sorted(set(sa) ^ set(sb))
    out = []
    for key in sa:
        if not same_bits(sa[key], sb[key]):
            d = np.abs(sa[key].astype(np.float64) - sb[key].astype(np.float64))
            out.append((key, int(np.count_nonzero(sa[key] != sb[key])), sa[key].size, float(np.nanmax(d))))
    return out


CIFAR_CASES = [
    # (name, depth, bits, tree)                        BASELINE config
    ("resnet20_8bit_admm", 20, 8, "admm"),           # configs[1]: the headline
    ("resnet20_2bit_admm", 20, 2, "admm"),           # configs[2] per rank
    ("resnet56_4bit_admm", 56, 4, "admm"),           # configs[3] per rank
    ("resnet20_8bit_cdf", 20, 8, "cdf"),             # configs[0]
]


@pytest.mark.parametrize("name,depth,bits,tree", CIFAR_CASES, ids=[c[0] for c in CIFAR_CASES])
def test_graph_replay_equals_eager_bit_for_bit_cifar(dev, name, depth, bits, tree):
    """bench.py's step (TrainStep(channels_last=True, qconv=True, fuse_bn=True, pack_bins=True), batch 128): five eager iterations
    against capture(warmup=3) + two replays from the same initial state and the same batches."""
    from alignq_amd import config
    from alignq_amd.resnet import resnet20_quant, resnet56_quant
    from alignq_amd.train_step import TrainStep
    old = (config.args.bitW, config.args.abitW, config.args.train_batch_size)
    config.args.bitW = config.args.abitW = bits
    config.args.train_batch_size = 128
    try:
        def make():
            torch.manual_seed(7)
            return (resnet20_quant if depth == 20 else resnet56_quant)(bits, bits, tree=tree).to(dev).train()
        g = torch.Generator().manual_seed(13)
        x = torch.randn(128, 3, 32, 32, generator=g).to(dev)
        y = torch.randint(0, 10, (128,), generator=g).to(dev)
        n_total, warm = 5, 3
        m1, m2 = make(), make()
        s1 = TrainStep(m1, channels_last=True, qconv=True, fuse_bn=True)
        s2 = TrainStep(m2, channels_last=True, qconv=True, fuse_bn=True)
        assert bool(s1.admms) == (tree == "admm")
        for _ in range(n_total):
            o1 = s1(x, y)
        s2.capture(x, y, warmup=warm)
        assert s2._graph is not None and s2._graph2 is None          # ONE graph: the form bench.py times at N = 1
        for _ in range(n_total - warm):
            o2 = s2(x, y)
        torch.cuda.synchronize()
        assert torch.isfinite(o1[1]) and torch.isfinite(o2[1])
        bad = differing(full_state(m1, s1, s1.admms), full_state(m2, s2, s2.admms))
        for k_, (a, b) in enumerate(zip(o1, o2)):
            if torch.is_tensor(a) and not same_bits(npy(a), npy(b)):
                bad.append(("output:%d" % k_, -1, a.numel(), float(np.abs(npy(a) - npy(b)).max())))
        assert not bad, "graph replay differs from eager in %d tensors, first: %s" % (len(bad), bad[:6])
    finally:
        config.args.bitW, config.args.abitW, config.args.train_batch_size = old


def test_graph_replay_equals_eager_bit_for_bit_office(dev):
    """BASELINE configs[4] at its real size (28 + 28 images of 224 x 224, OfficeTrainStep(channels_last=True) with its defaults, as
    bench.py --model resnet50_dann builds it): four eager iterations against capture(warmup=2) + two replays.  Every parameter
    behind this repository's kernels bit for bit; the stem's convolution and batch-norm sit behind torch's max-pool backward (atomic
    adds): compared to rounding, as in test_office_iteration_is_reproducible_run_to_run."""
    import alignq_amd.quantization  # noqa: F401
    from alignq_amd import config
    from alignq_amd.resnet_office import resnet50_dann
    from alignq_amd.train_step import OfficeTrainStep
    old = (config.args.bitW, config.args.abitW, config.args.train_batch_size, config.args.eval_batch_size)
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = config.args.eval_batch_size = 28
    try:
        B = 28
        g = torch.Generator().manual_seed(11)
        xs = torch.randn(B, 3, 224, 224, generator=g).to(dev)
        xt = torch.randn(B, 3, 224, 224, generator=g).to(dev)
        ys = torch.randint(0, 31, (B,), generator=g).to(dev)

        def make():
            return det_init_(resnet50_dann(8, 8)).to(dev).train()
        n_total, warm = 4, 2
        m1, m2 = make(), make()
        s1 = OfficeTrainStep(m1, lr=4e-5, channels_last=True)
        s2 = OfficeTrainStep(m2, lr=4e-5, channels_last=True)
        for _ in range(n_total):
            o1 = s1(xs, ys, xt)
        s2.capture(xs, ys, xt, warmup=warm)
        assert s2._graph is not None and s2._graph2 is None
        for _ in range(n_total - warm):
            o2 = s2(xs, ys, xt)
        torch.cuda.synchronize()
        assert torch.isfinite(o1[1]) and torch.isfinite(o2[1])
        a1, a2 = [b.admm0 for b in s1.blocks], [b.admm0 for b in s2.blocks]
        st1, st2 = full_state(m1, s1, a1), full_state(m2, s2, a2)
        stem = ("feature.conv1.", "feature.bn1.")
        loose = [key for key in st1 if key.split(":", 1)[1].startswith(stem)]
        for key in loose:
            np.testing.assert_allclose(st1[key], st2[key], rtol=1e-5, atol=1e-7 * float(np.abs(st1[key]).max()) + 1e-12, err_msg=key)
            st1.pop(key), st2.pop(key)
        bad = differing(st1, st2)
        assert same_bits(npy(o1[1]), npy(o2[1])) and same_bits(npy(o1[2]), npy(o2[2])), (float(o1[1]), float(o2[1]))
        assert not bad, "graph replay differs from eager in %d tensors, first: %s" % (len(bad), bad[:6])
    finally:
        config.args.bitW, config.args.abitW, config.args.train_batch_size, config.args.eval_batch_size = old
