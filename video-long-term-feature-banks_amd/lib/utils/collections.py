"""`AttrDict`: the dict-with-attribute-access the config tree is made of
(same behaviour as the reference's lib/utils/collections.py:22-35)."""


class AttrDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in self.__dict__:
            self.__dict__[name] = value
        else:
            self[name] = value
