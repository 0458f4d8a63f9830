chem_data = {}


class Specie:
    pass


def get_node_attributes(*a, **k):
    raise NotImplementedError
