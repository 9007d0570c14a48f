class Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if hasattr(v, "to"):
                setattr(self, k, v.to(device))
        return self


class Batch(Data):
    pass
