"""Name -> class registries: the reference's plugin API for this path
(MQ/libs/modeling/models.py:1-50).  Same decorator / builder names and call shapes, so
`make_meta_arch('LocPointTransformer', **cfg['model'])` (train_cl.py:135) resolves to the
MI355X-backed classes in this package."""

backbones, necks, generators, meta_archs = {}, {}, {}, {}


def _register(table):
    def register(name):
        def deco(cls):
            table[name] = cls
            return cls
        return deco
    return register


register_backbone = _register(backbones)
register_neck = _register(necks)
register_generator = _register(generators)
register_meta_arch = _register(meta_archs)


def make_backbone(name, **kwargs):
    return backbones[name](**kwargs)


def make_neck(name, **kwargs):
    return necks[name](**kwargs)


def make_generator(name, **kwargs):
    return generators[name](**kwargs)


def make_meta_arch(name, **kwargs):
    return meta_archs[name](**kwargs)
