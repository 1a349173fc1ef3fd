"""End-to-end drop-in runs on the GPU: `NeuRec.properties` + `conf/*.properties` + a `.rating`
file -> `neurec_amd.main` -> logs and metric lines in the reference's format, with the
evaluator's two entrances (on-device factor path, plugin score-matrix path) agreeing with the
CPU statement of the reference evaluator."""
import os
import re

import numpy as np
import pytest

from neurec_amd import defaults

pytestmark = pytest.mark.gpu


def _write_dataset(root, n_users=120, n_items=90, seed=3):
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "dataset"), exist_ok=True)
    with open(os.path.join(root, "dataset", "toy.rating"), "w") as f:
        for u in range(n_users):
            liked = (u % 6) * 15 + rng.choice(15, 10, replace=False)       # 6 taste clusters
            for it in liked:
                f.write("%d\t%d\t%d\t%d\n" % (u + 7, it + 300, 5, rng.randint(1, 10**6)))


def _run(tmp_path, argv):
    from neurec_amd.main import main
    path = defaults.write_default_configs(str(tmp_path), overrides={
        "data.input.path": os.path.join(str(tmp_path), "dataset"), "data.input.dataset": "toy",
        "test_batch_size": "64"})
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        return main(argv=argv, properties=path)
    finally:
        os.chdir(cwd)


def _log_text(tmp_path, model):
    folder = os.path.join(str(tmp_path), "log", "toy", model)
    files = os.listdir(folder)
    assert len(files) == 1 and files[0].startswith("toy_%s_" % model) and files[0].endswith(".log")
    with open(os.path.join(folder, files[0])) as f:
        return f.read()


def _oracle_line(model, evaluator):
    """The reference driver restated on the CPU: np.matmul-free fmaf scores -> -inf mask ->
    cpp_evaluate_matrix semantics -> float32 mean -> '%.8f' string."""
    from oracle import native
    P, Q = [t.cpu().numpy() for t in model.get_eval_factors()]
    uni = evaluator.evaluator
    users = list(uni.user_pos_test.keys())
    S = native.score_gemm(P, np.asarray(users, np.int32), Q)
    for r, u in enumerate(users):
        if u in uni.user_pos_train and len(uni.user_pos_train[u]) > 0:
            S[r][uni.user_pos_train[u]] = -np.inf                            # uni_evaluator.py:140-143
    res = native.eval_matrix(S, [uni.user_pos_test[u] for u in users], uni.metrics, uni.max_top)
    final = np.mean(res, axis=0).reshape(uni.metrics_num, uni.max_top)[:, uni.top_show - 1].reshape(-1)
    return "\t".join([("%.8f" % x).ljust(12) for x in final])


def test_mf_config_drops_in_and_learns(tmp_path):
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    model = _run(tmp_path, ["--recommender=MF", "--epochs=12", "--batch_size=128",
                            "--learning_rate=0.01", "--reg_mf=0.001", "--verbose=4"])
    text = _log_text(tmp_path, "MF")
    assert "Dataset name: toy" in text and "MF's hyperparameters:" in text
    assert "metrics:\tPrecision@10" in text
    iters = re.findall(r"\[iter (\d+) : loss : ([0-9.]+), time: ([0-9.]+)\]", text)
    assert [int(i[0]) for i in iters] == list(range(1, 13))
    losses = [float(i[1]) for i in iters]
    assert losses[-1] < 0.8 * losses[0]                                      # BPR loss goes down
    evals = re.findall(r"epoch (\d+):\t(.+)", text)
    assert [int(e[0]) for e in evals] == [4, 8, 12]                          # every `verbose` epochs
    last = evals[-1][1].split("\t")
    assert len(last) == 10 and all(re.fullmatch(r"\d\.\d{8}\s*", x) for x in last)
    # factor path == score-matrix (plugin) path == CPU oracle, to the last printed digit
    uni = model.evaluator.evaluator
    users = list(uni.user_pos_test.keys())
    line_factor = uni._format(uni._evaluate_factors(model, users))
    line_scores = uni._format(uni._evaluate_scores(model, users))
    assert line_factor == line_scores == _oracle_line(model, model.evaluator)
    assert float(line_factor.split("\t")[4]) > 0.3                           # NDCG@10 is the 5th number
    # predict contract: [B, I] float32 array; candidate mode -> list of per-user arrays
    full = model.predict([0, 5, 9], None)
    assert full.shape == (3, model.num_items) and full.dtype == np.float32
    cand = model.predict([0, 5], [[1, 2, 3], [7]])
    assert [len(c) for c in cand] == [3, 1] and np.array_equal(cand[0], full[0][[1, 2, 3]])


def test_lightgcn_config_drops_in(tmp_path):
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    model = _run(tmp_path, ["--recommender=LightGCN", "--epochs=6", "--batch_size=256",
                            "--n_layers=3", "--topk=[5,10]", "--metric=[\"Recall\",\"NDCG\"]"])
    text = _log_text(tmp_path, "LightGCN")
    assert "use the pre adjcency matrix" not in text          # that line goes to stdout, as in the reference
    evals = re.findall(r"epoch (\d+):\t(.+)", text)
    assert [int(e[0]) for e in evals] == list(range(6))                      # LightGCN.py:172-180
    assert "metrics:\tRecall@5" in text and "NDCG@10" in text
    first, last = [float(evals[i][1].split("\t")[3]) for i in (0, -1)]
    assert first > 0.3 and last > 0.3                                        # cluster structure is recovered
    model._final = None
    assert evals[-1][1] == _oracle_line(model, model.evaluator)


def test_lightgcn_embed_size_beyond_the_scoring_loop(tmp_path):
    """embed_size 160: the table runs zero-padded at 256 columns and the evaluation scores through the
    wide GEMM (engine.ScoreGemmWide) — the logged line equals the CPU restatement of the reference evaluator."""
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    model = _run(tmp_path, ["--recommender=LightGCN", "--epochs=2", "--batch_size=256", "--embed_size=160",
                            "--n_layers=2", "--topk=[5,10]", "--metric=[\"Recall\",\"NDCG\"]"])
    assert model.engine.d == 256 and model.engine.d_real == 160
    evals = re.findall(r"epoch (\d+):\t(.+)", _log_text(tmp_path, "LightGCN"))
    assert [int(e[0]) for e in evals] == [0, 1]
    model._final = None
    assert evals[-1][1] == _oracle_line(model, model.evaluator)


def test_candidate_negative_mode_and_groups(tmp_path):
    """rec.evaluate.neg > 0 (leave-one-out protocol) and group_view go through the same kernels."""
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    model = _run(tmp_path, ["--recommender=MF", "--epochs=2", "--batch_size=128", "--splitter=loo",
                            "--rec.evaluate.neg=20", "--topk=[5]", "--verbose=2"])
    text = _log_text(tmp_path, "MF")
    line = re.findall(r"epoch 2:\t(.+)", text)[0].split("\t")
    assert len(line) == 5
    # restate candidate-mode evaluation on the CPU (uni_evaluator.py:123-131)
    from oracle import native
    uni = model.evaluator.evaluator
    P, Q = [t.cpu().numpy() for t in model.get_eval_factors()]
    users = list(uni.user_pos_test.keys())
    rows, truth = [], []
    for u in users:
        cand = list(uni.user_pos_test[u]) + uni.user_neg_test[u]
        rows.append(native.score_gemm(P, np.asarray([u], np.int32), Q)[0][cand])
        truth.append(list(range(len(uni.user_pos_test[u]))))
    width = max(len(r) for r in rows)
    S = np.full((len(rows), width), -np.inf, np.float32)
    for r, v in enumerate(rows):
        S[r, :len(v)] = v
    res = native.eval_matrix(S, truth, uni.metrics, uni.max_top)
    want = np.mean(res, axis=0).reshape(5, 5)[:, 4]
    assert [x.strip() for x in line] == ["%.8f" % x for x in want]


def test_ngcf_config_drops_in(tmp_path):
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    model = _run(tmp_path, ["--recommender=NGCF", "--epochs=8", "--batch_size=128",
                            "--learning_rate=0.01", "--verbose=4", "--mess_dropout_ratio=0.0"])
    text = _log_text(tmp_path, "NGCF")
    assert "use the normalized adjacency matrix" in text and "using xavier initialization" in text
    iters = re.findall(r"\[iter (\d+) : loss : ([0-9.]+), time: ([0-9.]+)\]", text)
    assert len(iters) == 8 and float(iters[-1][1]) < float(iters[0][1])
    evals = re.findall(r"epoch (\d+):\t(.+)", text)
    assert [int(e[0]) for e in evals] == [4, 8]
    # with dropout off the evaluation forward is deterministic: the logged line is reproducible
    assert evals[-1][1] == _oracle_line(model, model.evaluator)
    assert model.get_eval_factors()[0].shape[1] == 48                        # concat of 3 blocks of 16


@pytest.mark.parametrize("emb,layers,width", [(64, "[64,64,64]", 256), (24, "[32,8]", 64)])
def test_ngcf_other_widths_drop_in(tmp_path, emb, layers, width):
    """embedding_size / layer_size other than the shipped 16 / [16, 16] (64 / [64, 64, 64]: the NGCF paper's) through
    the same plugin on the width-generic engine: log lines, loss falls, the logged evaluation line is the oracle's —
    at 256 concatenated columns through the general-GEMM scoring (no tile-maxima form at that width)."""
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    model = _run(tmp_path, ["--recommender=NGCF", "--epochs=8", "--batch_size=128", "--learning_rate=0.005",
                            "--verbose=4", "--mess_dropout_ratio=0.0", "--embedding_size=%d" % emb,
                            "--layer_size=" + layers])
    text = _log_text(tmp_path, "NGCF")
    assert "width-generic NGCF engine" in text
    iters = re.findall(r"\[iter (\d+) : loss : ([0-9.]+), time: ([0-9.]+)\]", text)
    assert len(iters) == 8 and float(iters[-1][1]) < float(iters[0][1])
    evals = re.findall(r"epoch (\d+):\t(.+)", text)
    assert [int(e[0]) for e in evals] == [4, 8]
    assert evals[-1][1] == _oracle_line(model, model.evaluator)
    assert model.get_eval_factors()[0].shape[1] == width


@pytest.mark.parametrize("extra,width", [(["--alg_type=gcn", "--learner=rmsprop", "--learning_rate=0.002"], 48),
                                         (["--alg_type=gcmc", "--learner=adagrad", "--learning_rate=0.05"], 32),
                                         (["--node_dropout_flag=True", "--node_dropout_ratio=0.2", "--learner=momentum",
                                           "--learning_rate=0.002"], 48),
                                         (["--learner=gd", "--learning_rate=0.01"], 48)])
def test_ngcf_conf_surface_drops_in(tmp_path, extra, width):
    """r05: conf/NGCF.properties' other settings through the plugin — alg_type gcn / gcmc, node dropout, the other
    learners of util/learner.py (the shipped width stays on the fused engine for a plain learner change; everything
    else runs on the width-generic engine): log lines, the loss falls, evaluation lines are printed; gcmc concatenates
    its dense layers only (NGCF.py:226-248: 2 x 16 columns)."""
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    model = _run(tmp_path, ["--recommender=NGCF", "--epochs=8", "--batch_size=128", "--verbose=4",
                            "--mess_dropout_ratio=0.0"] + extra)
    text = _log_text(tmp_path, "NGCF")
    iters = re.findall(r"\[iter (\d+) : loss : ([0-9.]+), time: ([0-9.]+)\]", text)
    assert len(iters) == 8 and float(iters[-1][1]) < float(iters[0][1])
    evals = re.findall(r"epoch (\d+):\t(.+)", text)
    assert [int(e[0]) for e in evals] == [4, 8]
    assert ("width-generic NGCF engine" in text) == (extra[0] != "--learner=gd")
    assert model.get_eval_factors()[0].shape[1] == width
    if "--node_dropout_flag=True" not in extra:              # (node dropout draws again at evaluation, NGCF.py:140-141)
        assert evals[-1][1] == _oracle_line(model, model.evaluator)


@pytest.mark.parametrize("learner,lr", [("rmsprop", "0.001"), ("momentum", "0.01")])
def test_multivae_other_learners_drop_in(tmp_path, learner, lr):
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    _run(tmp_path, ["--recommender=MultiVAE", "--epochs=30", "--batch_size=32", "--learning_rate=" + lr,
                    "--verbose=30", "--learner=" + learner])
    text = _log_text(tmp_path, "MultiVAE")
    iters = re.findall(r"\[iter (\d+) : loss : ([0-9.]+), time: ([0-9.]+)\]", text)
    assert len(iters) == 30 and float(iters[-1][1]) < float(iters[0][1])


def test_multivae_config_drops_in(tmp_path):
    """conf/MultiVAE.properties -> HIP Mult-VAE: log lines, loss falls; `predict` scores each user
    on their own history by default, and with reference_predict_rows=True on the reference's
    accumulating input row (MultiVAE.py:186-206) — both checked against the oracle forward."""
    from oracle import train
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    model = _run(tmp_path, ["--recommender=MultiVAE", "--epochs=30", "--batch_size=32",
                            "--learning_rate=0.01", "--verbose=15", "--total_anneal_steps=50",
                            "--reference_predict_rows=True"])
    text = _log_text(tmp_path, "MultiVAE")
    assert "reference_predict_rows=True" in text and model.predict_accumulates_rows
    iters = re.findall(r"\[iter (\d+) : loss : ([0-9.]+), time: ([0-9.]+)\]", text)
    assert len(iters) == 30 and float(iters[-1][1]) < float(iters[0][1])
    evals = re.findall(r"epoch (\d+):\t(.+)", text)
    assert [int(e[0]) for e in evals] == [15, 30]
    # the six taste clusters are learnable, even through the accumulating predict rows
    ndcg20 = float(evals[-1][1].split()[5])
    assert ndcg20 > 0.15
    users = [5, 17, 3, 44]
    got = model.predict(users, None).cpu().numpy()
    P = {k: v.cpu().numpy().astype(np.float64) for k, v in model.engine.P.items()}
    R = model.dataset.train_matrix.tocsr()
    X = np.zeros((len(users), model.num_items))
    row = np.zeros(model.num_items)
    for k, u in enumerate(users):
        row[R[u].indices] = 1
        X[k] = row
    want, _, _, _ = train.multivae_forward(X, [P["Wq0"], P["Wq1"]], [P["bq0"], P["bq1"]],
                                           [P["Wp0"], P["Wp1t"].T], [P["bp0"], P["bp1"]],
                                           np.ones_like(X), 1.0, np.zeros((len(users), 16)), 0.0, "tanh")
    assert np.abs(got - want).max() < 1e-5
    model.predict_accumulates_rows = False
    solo = model.predict(users, None).cpu().numpy()
    assert np.abs(solo[0] - want[0]).max() < 1e-5 and np.abs(solo[1] - want[1]).max() > 1e-4
    cand = model.predict(users, [[1, 2, 3]] * 4)
    assert np.allclose(cand[2], solo[2][[1, 2, 3]])
    # per-user inputs have a factor form ([g1(u) | 1]·[W_p1 | b_p1] = the logits, bias as the last factor
    # column): the evaluator's on-GPU factor path prints the same line as the predict() path, and the
    # factor scores ARE predict()'s numbers
    uni = model.evaluator.evaluator
    test_users = list(uni.user_pos_test.keys())
    Pf, Qf = model.get_eval_factors()
    assert Pf.shape == (model.num_users, 33) and Qf.shape == (model.num_items, 33)
    from oracle import native
    S = native.score_gemm(Pf.cpu().numpy(), np.asarray(users, np.int32), Qf.cpu().numpy())
    np.testing.assert_array_equal(S, solo)
    line_factor = uni._format(uni._evaluate_factors(model, test_users))
    line_scores = uni._format(uni._evaluate_scores(model, test_users))
    assert line_factor == line_scores
    model.predict_accumulates_rows = True
    assert model.get_eval_factors() is None                  # the accumulating rows have no factor form


@pytest.mark.parametrize("p_dim", ["[200,600]", "[40]", "[8,24,48]"])
def test_multivae_any_p_dim_drops_in(tmp_path, p_dim):
    """conf/MultiVAE.properties:3's alternatives (p_dim = [200, 600], [200], ...) run through the same plugin on the
    width-generic engine: log lines, loss falls, predict() equals the oracle's forward of the trained weights."""
    from oracle import train
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    model = _run(tmp_path, ["--recommender=MultiVAE", "--epochs=30", "--batch_size=32", "--learning_rate=0.003",
                            "--verbose=15", "--total_anneal_steps=50", "--p_dim=" + p_dim])
    text = _log_text(tmp_path, "MultiVAE")
    assert "width-generic Mult-VAE engine" in text
    iters = re.findall(r"\[iter (\d+) : loss : ([0-9.]+), time: ([0-9.]+)\]", text)
    assert len(iters) == 30 and float(iters[-1][1]) < float(iters[0][1])
    evals = re.findall(r"epoch (\d+):\t(.+)", text)
    assert [int(e[0]) for e in evals] == [15, 30]
    assert float(evals[-1][1].split()[5]) > 0.15
    eng = model.engine
    f = lambda ts: [t.cpu().numpy().astype(np.float64) for t in ts]
    users = [5, 17, 3, 44]
    X = np.asarray(model.dataset.train_matrix.tocsr()[users].todense(), dtype=np.float64)
    _, _, _, want = train.multivae_general(X, f(eng.Wq), f(eng.bq), f(eng.Wp), f(eng.bp), np.ones_like(X), 1.0,
                                           np.zeros((len(users), eng.z)), 0.0, 0.0, "tanh", is_training=0.0,
                                           want_grads=False)
    got = model.predict(users, None).cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-5
    cand = model.predict(users, [[1, 2, 3]] * 4)
    assert np.allclose(cand[2], got[2][[1, 2, 3]])


def test_mf_pointwise_and_other_learners_drop_in(tmp_path):
    """conf/MF.properties with is_pairwise=False (PointwiseSampler, sigmoid cross-entropy) and a
    non-Adam learner: runs end to end, loss falls, evaluation line printed."""
    _write_dataset(str(tmp_path))
    np.random.seed(2018)
    _run(tmp_path, ["--recommender=MF", "--epochs=6", "--batch_size=256", "--learning_rate=0.05",
                    "--is_pairwise=False", "--loss_function=cross_entropy", "--learner=adagrad",
                    "--num_negatives=2", "--verbose=6", "--embedding_size=16"])
    text = _log_text(tmp_path, "MF")
    iters = re.findall(r"\[iter (\d+) : loss : ([0-9.]+), time: ([0-9.]+)\]", text)
    assert len(iters) == 6 and float(iters[-1][1]) < float(iters[0][1])
    assert len(re.findall(r"epoch 6:\t", text)) == 1
