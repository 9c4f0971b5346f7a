"""Content hash of a native library's inputs, kept in a sidecar `<lib>.buildhash` next to it.

__graft_entry__.build() rebuilds when the sidecar does not match sha256(sources + flags) — file times play no part (a checkout can
restore sources older than a stray .so) — and host/capi.py refuses to load a library that is stale with respect to the sources
it sits next to."""
import hashlib
import os


def digest(paths, flags=()):
    h = hashlib.sha256()
    for f in flags:
        h.update(f.encode() + b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def sidecar(lib):
    return lib + ".buildhash"


def read(lib):
    try:
        return open(sidecar(lib)).read().strip()
    except OSError:
        return None


def write(lib, value):
    with open(sidecar(lib), "w") as fh:
        fh.write(value + "\n")
