"""YAML config surface: the reference's drivers do ``munch.munchify(yaml.safe_load(open(path)))``
(registration/train_HOUV.py:140, test.py:85, test_mult.py:94).  ``munch`` is not a dependency here: ``Config`` is a
small attribute-dict with the same access patterns (``args.batch_size``, ``args['kernel']``, ``str(args)``), and it
accepts the reference's ``cfgs/*.yaml`` files unchanged."""
import yaml


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(x):
        if isinstance(x, dict):
            return Config({k: Config.wrap(v) for k, v in x.items()})
        if isinstance(x, (list, tuple)):
            return type(x)(Config.wrap(v) for v in x)
        return x


# keys of registration/cfgs/houv.yaml:1-40 that the HOUV drivers read, with the reference's values
DEFAULTS = dict(batch_size=100, workers=0, model_name="houv", load_model=None, work_dir="log/", flag="debug",
                manual_seed=2021, num_points=2048, max_angle=180, max_trans=0.5, benchmark="mvp", kernel=32,
                lr=0.01, l=0, r=4, combine=False)


def load_config(path):
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    cfg = Config.wrap({**DEFAULTS, **raw})
    return cfg
