"""`ouster.sdk.examples.reference` is the reference's own python/src/ouster/sdk/examples/reference.py, staged next to the
staged tests (oracle/_ref/pytests/examples): it is the independent formula those tests compare against."""
import os

_staged = os.path.join(os.path.dirname(os.path.abspath(__file__)), *[".."] * 5, "oracle", "_ref", "pytests", "examples")
__path__.append(os.path.normpath(_staged))
