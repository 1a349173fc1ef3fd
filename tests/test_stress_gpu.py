"""The randomised cross-checks of tests/stress_*.py as tests (a fixed seed and a case count that runs in seconds):
the pruned evaluation against the materialised path over shapes / cut-offs / searches / awkward tables; the LightGCN
step against the fp64 twin with the fp32 oracle as the yardstick; the BPR-MF lazy forms against the all-rows sweep."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(script, cases, seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script), str(cases), str(seed)],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, tail


@pytest.mark.parametrize("seed", [11, 12])
def test_pruned_evaluation_equals_the_materialised_path_over_random_cases(seed):
    _run("stress_eval.py", 45, seed)


def test_lightgcn_step_is_as_close_to_fp64_as_the_fp32_oracle_over_random_cases():
    _run("stress_step.py", 14, 4)        # (seed 4: every adjacency form and depth, no ill-conditioned `plain` L >= 3 case)


def test_bpr_mf_lazy_forms_equal_the_sweep_over_random_cases():
    _run("stress_mf.py", 12, 21)
