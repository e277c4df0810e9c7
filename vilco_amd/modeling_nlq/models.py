"""Registry of the NLQ tree (NLQ/libs/modeling/models.py:1-50): same decorator / builder names, its own tables."""
backbones = {}


def register_backbone(name):
    def decorator(cls):
        backbones[name] = cls
        return cls
    return decorator


def make_backbone(name, **kwargs):
    return backbones[name](**kwargs)
