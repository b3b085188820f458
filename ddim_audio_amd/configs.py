"""Config objects in the shape the reference passes to ``Model(config)``.

The reference builds a nested ``argparse.Namespace`` from ``configs/audio.yml``
(reference ``main.py:19-203``, ``utils.py:4-12``) and ``Model`` reads
``config.model.{ch,krn,res,channels,f_size,dtype,transformers.*}`` plus
``config.diffusion.num_diffusion_timesteps`` (reference ``models/diffusion.py:170-235``).
The values below are the hyper-parameters of ``configs/audio.yml:1-109`` restated as data.
"""
import argparse
import copy


def dict2namespace(d):
    """Recursive dict -> Namespace (same contract as reference ``utils.py:4-12``)."""
    ns = argparse.Namespace()
    for k, v in d.items():
        setattr(ns, k, dict2namespace(v) if isinstance(v, dict) else v)
    return ns


_AUDIO = {
    "model": {
        "dtype": "torch.cuda.FloatTensor",
        "type": "simple",
        "transformers": {
            "imports": "import transformers; from transformers.models.fnet.modeling_fnet import FNetEncoder",
            "module": "FNetEncoder",
            "config": "transformers.FNetConfig",
            "kwargs": {
                "hidden_size": 512,
                "num_hidden_layers": 12,
                "intermediate_size": 2048,
                "hidden_act": "gelu_new",
                "hidden_dropout_prob": 0.1,
                "initializer_range": 0.02,
                "layer_norm_eps": 0.000001,
            },
            "channels": 512,
            "dtype": "torch.cuda.FloatTensor",
        },
        "channels": 2,
        "t_size": 1024,
        "f_size": 256,
        "ch": [32, 64, 96, 128, 192, 256],
        "krn": [3, 3, 3, 3, 3, 3],
        "res": [2, 2, 3, 3, 3, 3],
        "var_type": "fixedlarge",
        "ema_rate": 0.9999,
        "ema": True,
    },
    "diffusion": {
        "beta_schedule": "linear",
        "beta_start": 0.0001,
        "beta_end": 0.02,
        "num_diffusion_timesteps": 1000,
    },
    "training": {"batch_size": 14, "n_iters": 5000000, "snapshot_freq": 5000},
    "sampling": {"batch_size": 64, "last_only": True, "num_samples": 2, "t_size": 8192},
    "optimization": {
        "optimizer": {
            "transformer": {
                "top_level_name": ["transformer"],
                "weight_decay": 0.0001,
                "optimizer": "AdamW",
                "warmup": 10000,
                "lr": 0.0005,
                "beta": [0.9, 0.998],
                "amsgrad": False,
                "eps": 0.000001,
            },
            "default": {
                "top_level_name": [],
                "weight_decay": 0.00001,
                "optimizer": "AdaBelief",
                "warmup": 1000,
                "lr": 0.0003,
                "beta": [0.9, 0.999],
                "amsgrad": False,
                "eps": 0.00000001,
                "clip_step": None,
                "norm_ord": 2,
            },
        },
        "grad_norm": {
            "transformer": {"top_level_name": [], "grad_clip": 1},
            "default": {"top_level_name": [], "grad_clip": 1},
        },
    },
}


def audio_dict(dtype="torch.cuda.FloatTensor", fnet_dtype=None):
    d = copy.deepcopy(_AUDIO)
    d["model"]["dtype"] = dtype
    d["model"]["transformers"]["dtype"] = fnet_dtype if fnet_dtype is not None else dtype
    return d


def audio_config(dtype="torch.cuda.FloatTensor", fnet_dtype=None):
    """The full-size network of ``configs/audio.yml`` (47,155,266 parameters)."""
    return dict2namespace(audio_dict(dtype, fnet_dtype))


def tiny_dict(dtype="torch.cuda.FloatTensor", fnet_dtype=None):
    """A small network with the same topology rules (verified constructible).

    Channel widths are the first three of audio.yml (kernels are instantiated per width pair);
    f_size 32 with three levels gives an FNet token width of 96 * (32 / 4) = 768.
    """
    d = audio_dict(dtype, fnet_dtype)
    m = d["model"]
    m["ch"] = [32, 64, 96]
    m["krn"] = [3, 3, 3]
    m["res"] = [1, 2, 1]
    m["f_size"] = 32
    m["t_size"] = 16
    m["transformers"]["kwargs"].update(hidden_size=64, num_hidden_layers=2, intermediate_size=128)
    m["transformers"]["channels"] = 64
    return d


def tiny_config(dtype="torch.cuda.FloatTensor", fnet_dtype=None):
    return dict2namespace(tiny_dict(dtype, fnet_dtype))


def micro_dict(dtype="torch.cuda.FloatTensor", fnet_dtype=None):
    """The smallest network the kernels are instantiated for (two levels, one block each, one FNet layer): used for the
    reference-written checkpoint fixture (tests/golden/ckpt_micro.pth, about 0.7 M parameters).  The optimizer groups are
    listed default-first so that the reference's checkpoint -- which keeps only the LAST optimizer of its dict
    (runners/diffusion.py:162-189) -- carries the small transformer group."""
    d = audio_dict(dtype, fnet_dtype)
    m = d["model"]
    m["ch"] = [32, 64]
    m["krn"] = [3, 3]
    m["res"] = [1, 1]
    m["f_size"] = 8
    m["t_size"] = 8
    m["transformers"]["kwargs"].update(hidden_size=32, num_hidden_layers=1, intermediate_size=64)
    m["transformers"]["channels"] = 32
    o = d["optimization"]["optimizer"]
    d["optimization"]["optimizer"] = {"default": o["default"], "transformer": o["transformer"]}
    d["optimization"]["optimizer"]["default"]["optimizer"] = "Adam"  # AdaBelief's source is absent upstream (un-vendored submodule)
    return d


def micro_config(dtype="torch.cuda.FloatTensor", fnet_dtype=None):
    return dict2namespace(micro_dict(dtype, fnet_dtype))
