def make_grid(*a, **k):
    raise NotImplementedError("stub")
