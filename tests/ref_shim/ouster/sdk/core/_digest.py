"""`ouster.sdk.core._digest`: the REFERENCE's own module (python/src/ouster/sdk/core/_digest.py: md5 digests of packet fields
and frame planes, and their comparison with the *_digest.json fixtures), executed verbatim from where oracle/Makefile staged it
(oracle/_ref/pytests/core/_digest.py, git-ignored).  Without the staged file only the JSON loader exists."""
import json
import os

_staged = os.path.join(os.environ.get("OUSTER_REF_STAGED", ""), "core", "_digest.py")
if os.path.exists(_staged):
    with open(_staged) as _f:
        exec(compile(_f.read(), _staged, "exec"), globals())
else:
    class StreamDigest:
        def __init__(self, data):
            self.data = data

        @classmethod
        def from_json(cls, text):
            return cls(json.loads(text))
