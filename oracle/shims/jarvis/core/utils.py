def random_colors(*a, **k):
    raise NotImplementedError
