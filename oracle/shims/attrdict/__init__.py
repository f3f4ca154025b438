"""TEST INFRASTRUCTURE (oracle shim): minimal attrdict.AttrDict (attribute access on nested dicts),
as used by src/pipeline_config.py:33 and src/utils.py:133."""


class AttrDict(dict):
    def __getattr__(self, name):
        try:
            v = self[name]
        except KeyError:
            raise AttributeError(name)
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        return v

    def __setattr__(self, name, value):
        self[name] = value

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
            dict.__setitem__(self, key, v)
        return v
