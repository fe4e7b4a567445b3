"""`bs segment` configuration surface.

Behavioural mirror of /root/reference/bootstrapper/segment.py:10-163: method DEFAULTS,
parameter precedence DEFAULTS < TOML `<method>_params` < `-p key=value`, coordinate parsing,
the blockwise requirements, and the dispatch to the `ws` / `mws` / `cc` drivers under post/.
Pinned by tests/golden/host_cases.json.
"""
from ast import literal_eval

try:  # python >= 3.11
    import tomllib as _toml
except ImportError:  # pragma: no cover
    import tomli as _toml

_AFF_NBHD = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -9, 0], [0, 0, -9], [-3, 0, 0], [0, -27, 0], [0, 0, -27]]

DEFAULTS = {
    "ws": {
        "fragments_in_xy": True, "min_seed_distance": 10, "seed_eps": None, "epsilon_agglomerate": 0.0,
        "filter_fragments": 0.1, "remove_debris": 64, "thresholds": [0.2, 0.35, 0.5], "merge_function": "mean",
        "sigma": None, "noise_eps": None, "bias": None,
    },
    "mws": {
        "aff_neighborhood": _AFF_NBHD, "bias": [-0.4] * 3 + [-0.7] * 6, "sigma": None, "noise_eps": 0.001,
        "strides": [[1, 1, 1]] * 3 + [[2, 9, 9]] * 3 + [[3, 27, 27]] * 3, "randomized_strides": True,
        "filter_fragments": 0.1, "remove_debris": 64, "min_seed_distance": None, "global_bias": [1.0, -0.5],
    },
    "cc": {"threshold": 0.5, "sigma": None, "noise_eps": None, "remove_debris": 64},
}

_COORD_KEYS = ("roi_offset", "roi_shape", "block_shape", "context")


def parse_params(text):
    try:
        return literal_eval(text)
    except Exception:  # noqa: BLE001 - anything unparsable stays a string, like the reference
        return text


def parse_shape(value):
    """None / 'roi' pass through; strings split on spaces or commas; everything becomes ints."""
    if value is None or value == "roi":
        return value
    if isinstance(value, str):
        value = value.replace(",", " ").split()
    return [int(v) for v in value]


def get_method_params(method, params):
    out = {}
    for item in params:
        key, value = item.split("=")
        if key not in DEFAULTS[method]:
            raise ValueError(f"Invalid {method} parameter {key}")
        out[key] = parse_params(value)
    return out


def load_toml(path):
    with open(path, "rb") as f:
        return _toml.load(f)


def get_seg_config(config_file, method, **kwargs):
    config = load_toml(config_file)
    for key, value in kwargs.items():
        if key != "param" and value is not None:
            config["context" if key == "block_context" else key] = value
    params = {**DEFAULTS[method], **config.get(f"{method}_params", {}),
              **get_method_params(method, kwargs.get("param", ()))}
    for key in [k for k in config if k.endswith("_params")]:
        del config[key]
    for key in _COORD_KEYS:
        if key in config:
            config[key] = parse_shape(config[key])
    if config.get("blockwise", False):
        if method == "cc":
            raise ValueError("Blockwise connected components is not supported!")
        if "db" not in config:
            raise ValueError("Blockwise requires a database config!")
        if "lut_dir" not in config:
            config["lut_dir"] = config["seg_dataset_prefix"].replace("segmentations", "luts")
    return {**config, **params}


def run_segmentation(config_file, mode="ws", **kwargs):
    config = get_seg_config(config_file, mode, **kwargs)
    if mode == "ws":
        from .post.watershed import watershed_segmentation
        return watershed_segmentation(config)
    if mode == "cc":
        from .post.connected_components import cc_segmentation
        return cc_segmentation(config)
    if mode == "mws":
        from .post.watershed_mutex import mutex_watershed_segmentation
        return mutex_watershed_segmentation(config)
    raise ValueError(f"Unknown segmentation mode: {mode}")
