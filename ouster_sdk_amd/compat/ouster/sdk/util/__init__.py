"""ouster.sdk.util: metadata resolution next to a capture (python/src/ouster/sdk/util/metadata.py)."""
import os


def resolve_metadata(data_path):
    """The JSON file sharing the longest file-name prefix with the capture, in the same directory."""
    d, name = os.path.split(data_path)
    base = os.path.splitext(name)[0]
    best, best_len = None, 0
    for f in sorted(os.listdir(d or ".")):
        if not f.endswith(".json"):
            continue
        n = len(os.path.commonprefix([base, os.path.splitext(f)[0]]))
        if n > best_len:
            best, best_len = f, n
    return os.path.join(d, best) if best else None


def resolve_metadata_multi(data_path):
    m = resolve_metadata(data_path)
    return [m] if m else []
