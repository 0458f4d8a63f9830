"""Stand-in for jarvis.db.figshare (dataset downloads): importable, never usable offline."""


def data(*a, **k):
    raise NotImplementedError("jarvis.db.figshare.data needs the network; the oracle passes its own loaders")


def get_request_data(*a, **k):
    raise NotImplementedError
