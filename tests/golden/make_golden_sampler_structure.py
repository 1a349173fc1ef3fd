"""Golden vectors for the host-side structure builders of data/sampler.py, produced by the
REFERENCE's own functions: `_generate_positive_items` (data/sampler.py:24-39) and
`_generative_time_order_positive_items` (:42-68).  The functions are lifted out of the reference
source with `ast` (the module itself imports TensorFlow-dependent packages) and executed as
they are.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_sampler_structure.py
"""
import ast
import json
import os

import numpy as np

REF = "/root/reference/data/sampler.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def reference_functions(names):
    src = open(REF).read()
    tree = ast.parse(src)
    ns = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), REF, "exec"), ns)
    return [ns[n] for n in names]


def main():
    gen_pos, gen_time = reference_functions(["_generate_positive_items",
                                             "_generative_time_order_positive_items"])
    rng = np.random.RandomState(2018)
    users = rng.permutation(60)[:35].tolist()
    seqs = {int(u): rng.choice(200, int(rng.randint(1, 15)), replace=False).tolist() for u in users}
    cases = {"seqs": {str(k): v for k, v in seqs.items()}, "order": [int(u) for u in seqs]}
    upl, ul, pl = gen_pos(seqs)
    cases["positive"] = {"user_pos_len": upl, "users": ul, "pos": pl}
    for h in (1, 2, 4):
        upl, ul, rl, pl = gen_time(seqs, high_order=h)
        cases["time_%d" % h] = {"user_pos_len": upl, "users": ul, "recent": rl, "pos": pl}
    with open(os.path.join(OUT, "sampler_structure.json"), "w") as f:
        json.dump(cases, f)
    print("wrote sampler_structure.json", {k: len(v["users"]) for k, v in cases.items() if isinstance(v, dict) and "users" in v})


if __name__ == "__main__":
    main()
