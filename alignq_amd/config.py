"""Process-wide options the reference ops read from its argparse namespace `args`
(utils/options.py:32-95; utils/options_office.py:64-98).  Same names, same defaults; assign fields
(`alignq_amd.config.args.act_range = 2`) or replace the object with the reference's own `args`
via `use_args(namespace)` for a drop-in."""
from types import SimpleNamespace

args = SimpleNamespace(
    act_range=2.0,          # options.py: ACT_RANGE
    method="ours",          # options.py: METHOD
    bitW=8,
    abitW=8,
    lam=1.0,                # options.py: LAMBDA
    lam2=4.0,               # options.py: LAMBDA2
    train_batch_size=128,
    eval_batch_size=100,
    stage="second",
    gpus=[0],
    global_corr=None,       # not a reference option: True / a process group -> exact-global-batch corr
    pack_bins=False,        # not a reference option: plain quantiser nodes keep int8/int16 bins for backward
)


def use_args(namespace):
    """Adopt an external namespace (e.g. the reference's parsed `args`) as the live option object."""
    global args
    for k, v in vars(args).items():
        if not hasattr(namespace, k):
            setattr(namespace, k, v)
    args = namespace
    return args


def get():
    return args
