"""YAML config loading with attribute + item access (stands in for OmegaConf, which the reference uses at
indextts/infer.py:210; keys read on the path are listed in SURVEY.md §5)."""
import yaml


class Config(dict):
    """dict with attribute access, recursive; `cfg.gpt.max_mel_tokens`, `cfg["gpt"]`, `**cfg.gpt` all work."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = _wrap(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def get(self, k, default=None):
        return self[k] if k in self else default


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, Config):
        return Config(v)
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def load_config(path: str) -> Config:
    with open(path, "r", encoding="utf-8") as f:
        return Config(yaml.safe_load(f) or {})
