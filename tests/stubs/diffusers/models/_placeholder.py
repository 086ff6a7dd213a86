def placeholder(name):
    """a name the reference imports but never instantiates on the SD-v2 / GeoWizard path"""
    def __init__(self, *a, **k):
        raise NotImplementedError("tests/stubs: diffusers.%s is not on the E2E-FT path" % name)
    return type(name, (), {"__init__": __init__})
