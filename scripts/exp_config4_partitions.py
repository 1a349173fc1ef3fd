"""BASELINE configs[3] on ONE GPU: one rank's share of an 8-rank job under the row and the column partition
(bench.leg_config4_partitions; the driver line carries it under config4.partitions).
    python scripts/exp_config4_partitions.py [scale] [rows|cols|both]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import leg_config4_partitions  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
which = sys.argv[2] if len(sys.argv) > 2 else "both"
print("RESULT " + json.dumps(leg_config4_partitions(torch.device("cuda", 0), scale, which)))
