# Runtime defaults (same keys as the reference's configs/_base_/default_runtime.py).
weight = None
resume = False
evaluate = False
test_only = False
seed = None
save_path = "exp/default"
num_worker = 8
batch_size = 16
batch_size_val = None
batch_size_test = None
epoch = 100
eval_epoch = 100
sync_bn = False
enable_amp = False
empty_cache = False
find_unused_parameters = False
mix_prob = 0
param_dicts = None
hooks = [
    dict(type="CheckpointLoader"),
    dict(type="IterationTimer", warmup_iter=2),
    dict(type="InformationWriter"),
    dict(type="CheckpointSaver", save_freq=None),
]
train = dict(type="DefaultTrainer")
