"""Default configuration values, and a writer that materialises them as the ini files the
reference ships (`NeuRec.properties`, `conf/<Model>.properties`) for a fresh working directory.
Existing files written for the reference are read unchanged; this module only supplies
defaults when there are none."""
import os

LIBRARY = [
    ("recommender", "MF"), ("config_dir", "./conf"), ("gpu_id", "0"), ("gpu_mem", "0.5"),
    ("data.input.path", "dataset"), ("data.input.dataset", "ml-100k"),
    ("data.column.format", "UIRT"), ("data.convert.separator", "'\\t'"),
    ("user_min", "0"), ("item_min", "0"), ("splitter", "ratio"), ("ratio", "0.8"),
    ("by_time", "False"), ("metric", '["Precision", "Recall", "NDCG", "MAP", "MRR"]'),
    ("topk", "[10, 20]"), ("group_view", "None"), ("rec.evaluate.neg", "0"),
    ("test_batch_size", "128"), ("num_thread", "8"),
]

MODELS = {
    "MF": [("epochs", "300"), ("batch_size", "512"), ("embedding_size", "64"), ("reg_mf", "0.0"),
           ("learning_rate", "0.001"), ("learner", "adam"), ("num_negatives", "1"),
           ("is_pairwise", "True"), ("loss_function", "bpr"), ("init_method", "normal"),
           ("stddev", "0.01"), ("verbose", "1")],
    "LightGCN": [("lr", "0.01"), ("reg", "1e-3"), ("embed_size", "64"), ("n_layers", "6"),
                 ("batch_size", "1024"), ("epochs", "500"), ("n_fold", "100"),
                 ("adj_type", "pre")],
    "NGCF": [("epochs", "500"), ("batch_size", "512"), ("embedding_size", "16"),
             ("layer_size", "[16,16]"), ("learning_rate", "0.001"), ("node_dropout_flag", "False"),
             ("adj_type", "norm"), ("alg_type", "ngcf"), ("loss_function", "BPR"),
             ("learner", "adam"), ("reg", "0.0"), ("node_dropout_ratio", "0.1"),
             ("mess_dropout_ratio", "0.1"), ("embed_init_method", "xavier_normal"),
             ("weight_init_method", "xavier_normal"), ("stddev", "0.01"), ("verbose", "1")],
    "MultiVAE": [("epochs", "500"), ("p_dim", "[16,32]"), ("batch_size", "512"), ("reg", "0.0"),
                 ("learning_rate", "0.001"), ("activation", "tanh"),
                 ("loss_function", "multinominal-likelihood"), ("learner", "adam"),
                 ("anneal_cap", "0.2"), ("total_anneal_steps", "2000"),
                 ("weight_init_method", "xavier_normal"), ("bias_init_method", "tnormal"),
                 ("stddev", "0.01"), ("verbose", "1")],
}


def write_default_configs(directory, overrides=None, model_overrides=None):
    """Create NeuRec.properties and conf/*.properties under `directory`; returns the path of
    NeuRec.properties.  `overrides` / `model_overrides[model]` replace individual values."""
    os.makedirs(os.path.join(directory, "conf"), exist_ok=True)
    lib = dict(LIBRARY)
    lib.update(overrides or {})
    lib["config_dir"] = os.path.join(directory, "conf")
    path = os.path.join(directory, "NeuRec.properties")
    with open(path, "w") as f:
        f.write("[default]\n")
        for k, _ in LIBRARY:
            f.write("%s=%s\n" % (k, lib[k]))
        for k in lib:
            if k not in dict(LIBRARY):
                f.write("%s=%s\n" % (k, lib[k]))
    for model, items in MODELS.items():
        vals = dict(items)
        vals.update((model_overrides or {}).get(model, {}))
        with open(os.path.join(directory, "conf", model + ".properties"), "w") as f:
            f.write("[hyperparameters]\n")
            for k in vals:
                f.write("%s=%s\n" % (k, vals[k]))
    return path
