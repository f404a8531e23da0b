#!/usr/bin/env python3
"""Stage one of the reference's C++ test files (tests/*.cpp of ouster-sdk) for compilation against this repo's mirror of
the ouster_core API: the file is copied as it is, except that the TEST / TEST_P blocks named with --drop are taken out
whole (they exercise packet kinds SURVEY.md section 8 leaves out of scope: IMU and zone-monitoring packets) and a
comment takes their place.  Nothing else is touched; the output goes under the git-ignored oracle/_ref/.
Test infrastructure only (see oracle/Makefile, target cpptests)."""
import argparse
import re
import sys


def drop_block(lines, suite, name):
    head = re.compile(r"^TEST(_P|_F)?\(\s*%s\s*,\s*%s\s*\)" % (re.escape(suite), re.escape(name)))
    for i, ln in enumerate(lines):
        if head.match(ln):
            depth, j, seen = 0, i, False
            while j < len(lines):
                depth += lines[j].count("{") - lines[j].count("}")
                seen = seen or "{" in lines[j]
                if seen and depth == 0:
                    break
                j += 1
            if j == len(lines):
                raise SystemExit("unbalanced braces in %s.%s" % (suite, name))
            return lines[:i] + ["// [%s.%s is not staged: out of scope, see oracle/Makefile]\n" % (suite, name)] + lines[j + 1:], True
    return lines, False


def drop_function(lines, name):
    """A free function defined at file scope (its signature may span lines): from the line that starts its declaration
    to the closing brace."""
    for i, ln in enumerate(lines):
        if re.search(r"\b%s\s*\(" % re.escape(name), ln) and not ln.startswith((" ", "\t", "//", "TEST")):
            depth, j, seen = 0, i, False
            while j < len(lines):
                depth += lines[j].count("{") - lines[j].count("}")
                seen = seen or "{" in lines[j]
                if seen and depth == 0:
                    break
                j += 1
            if j == len(lines):
                raise SystemExit("unbalanced braces in %s" % name)
            return lines[:i] + ["// [function %s is not staged: see oracle/Makefile]\n" % name] + lines[j + 1:], True
    return lines, False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--drop", action="append", default=[], help="Suite.name of a TEST to leave out")
    ap.add_argument("--drop-function", action="append", default=[], help="file-scope helper function to leave out")
    a = ap.parse_args()
    lines = open(a.src).readlines()
    for d in a.drop:
        suite, name = d.split(".", 1)
        lines, found = drop_block(lines, suite, name)
        if not found:
            sys.exit("%s: no TEST %s" % (a.src, d))
    for d in a.drop_function:
        lines, found = drop_function(lines, d)
        if not found:
            sys.exit("%s: no function %s" % (a.src, d))
    open(a.dst, "w").writelines(lines)


if __name__ == "__main__":
    main()
