"""Entry point: `python -m neurec_amd.main [--key=value ...]` in a directory that holds
`NeuRec.properties` and `conf/<Model>.properties` — the reference's `python main.py`
(main.py:10-45): same seeds, same configuration files, same dispatch of
`recommender=<Name>` to `model/<family>_recommender/<Name>.py`, without a TensorFlow session.
"""
import importlib
import importlib.util
import os
import random
import sys

import numpy as np

_FAMILIES = ("general_recommender", "social_recommender", "sequential_recommender")


def find_recommender(name):
    """The class `name` from the first family package that has a module of that name."""
    for family in _FAMILIES:
        qualified = "neurec_amd.model.%s.%s" % (family, name)
        try:
            found = importlib.util.find_spec(qualified) is not None
        except ModuleNotFoundError:
            found = False
        if found:
            return getattr(importlib.import_module(qualified), name)
    raise ImportError("no recommender named '%s' under neurec_amd.model.{%s}" % (name, ",".join(_FAMILIES)))


def main(argv=None, properties="NeuRec.properties"):
    from .data.dataset import Dataset
    from .util import Configurator
    np.random.seed(2018)
    random.seed(2018)
    conf = Configurator(properties, default_section="hyperparameters", argv=argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # one rank of several (python -m torch.distributed.run --nproc-per-node N -m neurec_amd.main ...): every rank
        # takes the GPU of its LOCAL_RANK (parallel.init_from_env), `gpu_id` names a single device and is not used;
        # the plugins that have a multi-GPU form (LightGCN) ask parallel.get_comm() for the partition
        from . import parallel
        parallel.get_comm()
    else:
        os.environ["HIP_VISIBLE_DEVICES"] = str(conf["gpu_id"])     # main.py:17-18 (CUDA_VISIBLE_DEVICES)
    dataset = Dataset(conf)
    model = find_recommender(conf["recommender"])(None, dataset, conf)
    model.build_graph()
    model.train_model()
    return model


if __name__ == "__main__":
    main(sys.argv[1:])
