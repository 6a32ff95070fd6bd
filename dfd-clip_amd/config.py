"""Minimal attribute-access config node.

The reference builds its config tree from a fork of yacs (`CfgNode`), which is
not installable offline.  The model only needs three behaviours from it
(reference `src/models.py:97`, `:216`, `:251`, `:288`, `:310`, `:488`, `:494`):
attribute get/set, ``"key" in node`` and nested nodes.  A yacs `CfgNode` passed
in by a reference-style caller works unchanged because it offers the same three.
"""


class ConfigNode(dict):
    def __init__(self, init=None, new_allowed=True):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, ConfigNode):
            v = ConfigNode(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        out = ConfigNode()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, ConfigNode) else (list(v) if isinstance(v, list) else v)
        return out


def default_detector_config():
    """Same keys and defaults as `Detector.get_default_config` (reference `src/models.py:406-431`)."""
    C = ConfigNode()
    C.name = "Detector"
    C.foundation = "clip"
    C.architecture = "ViT-B/16"
    C.decode_mode = "stride"
    C.decode_stride = 2
    C.decode_indices = []
    C.out_dim = []
    C.losses = []
    C.concat_ref = 0
    C.adapter = ConfigNode()
    C.adapter.type = "none"
    C.train_mode = ConfigNode()
    C.op_mode = ConfigNode()
    C.op_mode.temporal_position = 1
    C.dropout = 0.0
    C.weight_decay = 0.01
    C.optimizer = "sgd"
    return C
