"""Registry of the NLQ tree (NLQ/libs/modeling/models.py:1-50): same decorator / builder names, its own tables."""
backbones, meta_archs = {}, {}


def register_backbone(name):
    def decorator(cls):
        backbones[name] = cls
        return cls
    return decorator


def make_backbone(name, **kwargs):
    return backbones[name](**kwargs)


def register_meta_arch(name):
    def decorator(cls):
        meta_archs[name] = cls
        return cls
    return decorator


def make_meta_arch(name, **kwargs):
    return meta_archs[name](**kwargs)
