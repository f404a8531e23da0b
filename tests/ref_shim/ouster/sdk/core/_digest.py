"""The part of ouster.sdk.core._digest the staged reference tests touch: loading a *_digest.json fixture."""
import json


class StreamDigest:
    def __init__(self, data):
        self.data = data

    @classmethod
    def from_json(cls, text):
        return cls(json.loads(text))
