class Atoms:  # placeholder; the oracle never builds jarvis Atoms
    pass


def get_supercell_dims(*a, **k):
    raise NotImplementedError
