"""Stand-in for jarvis.db.jsonutils: plain json load / dump (what the published helpers do)."""
import json


def loadjson(filename=""):
    with open(filename) as f:
        return json.load(f)


def dumpjson(data=[], filename=""):
    with open(filename, "w") as f:
        json.dump(data, f)
