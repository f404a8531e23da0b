#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/_ref): copy ONE function template out of a reference header into a git-ignored staging file
at build time, so that it is compiled as the reference wrote it without the rest of a header this image cannot compile
(impl/lidar_frame_impl.h needs Eigen::Tensor and the whole LidarFrame machinery).  The staged file is deleted after the
compile (oracle/Makefile); nothing of the reference's source enters the repository.
usage: stage_slice.py <header> <out> <first line of the declaration, after its template line>"""
import sys

src, out, decl = sys.argv[1], sys.argv[2], sys.argv[3]
lines = open(src).read().split("\n")
for i, ln in enumerate(lines):
    if ln.strip() == decl.strip() and i > 0 and lines[i - 1].startswith("template"):
        depth, j, seen = 0, i, False
        while True:
            depth += lines[j].count("{") - lines[j].count("}")
            seen |= "{" in lines[j]
            if seen and depth == 0:
                break
            j += 1
        body = "\n".join(lines[i - 1:j + 1])
        open(out, "w").write("// staged from %s:%d-%d at build time (oracle/stage_slice.py); not part of the repository\n%s\n"
                             % (src, i, j + 1, body))
        sys.exit(0)
sys.exit("declaration not found: " + decl)
