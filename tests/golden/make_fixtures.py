#!/usr/bin/env python3
"""How EVERYTHING under tests/golden/ is made (runs only where /root/reference is mounted).

The reference holds, for each regression case, the input deck (<case>.inp), the Plot3D
grid, for one case a point cloud of the initial state -- and, inside
testCases/regressionTests.py, the normalised L2 residuals its own binary must reproduce
after a fixed number of iterations.  This script

  * copies the DATA files of the twelve cases that lie inside the hot path
    (tests/golden/cases/<case>/) -- nothing of the reference's code is copied;
  * transcribes their truth vectors, iteration counts, ignored columns and the line of
    regressionTests.py each vector stands on (regression_truths.json: `truth`,
    `iterations`, `ignore`, `line`; the remaining keys of that file -- `_note`,
    `digits_exact`, `rtol` -- are this repository's annotations and are kept);
  * derives the one deck that is not a verbatim copy, uniformFlow.inp, by the edits
    listed in UNIFORMFLOW_EDITS (the reference runs that grid -- ten blocks joined with
    all eight patch orientations -- as a rans case whose truth is round-off by its own
    account, regressionTests.py:488-490; here the grid is run as an Euler LU-SGS deck).

  python tests/golden/make_fixtures.py            verify the committed tree against
                                                  what the reference yields (default)
  python tests/golden/make_fixtures.py --write    (re)generate the tree

Cases NOT taken: thermallyPerfect, supersonicMixing, dissociation (multi-species /
thermally perfect / chemistry) -- outside the path.
"""
import filecmp
import json
import os
import re
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
# case -> rank count whose truth vector applies (our blocks are never split, which is the
# 1-process decomposition; multiblockCylinder's two grid blocks are two blocks either way
# and the reference keeps one vector for it)
CASES = {
    "supersonicWedge": 1, "subsonicCylinder": 1, "multiblockCylinder": 2, "shockTube": 1,
    "viscousFlatPlate": 1, "couette": 1, "rae2822": 1, "turbFlatPlate": 1, "wallLaw": 1,
    "convectingVortex": 1, "transonicBump": 1,
    "uniformFlow": None,      # None: no truth taken (see above)
}
ITER_VARS = {"numIterations": 100, "numIterationsShort": 20}
UNIFORMFLOW_EDITS = [        # (regex, replacement) applied to the reference's deck
    (r"\r\n", "\n"),
    (r"equationSet: rans", "equationSet: euler"),
    (r"cflStart: 1000", "cflStart: 5"),
    (r"cflMax: 1000", "cflMax: 5"),
    (r"viscousFaceReconstruction: central\n", ""),
    (r"iterations: 1000", "iterations: 20"),
    (r"turbulenceModel: sst2003\n", ""),
]


def data_files(ref, case):
    """every data file of the case directory: deck, grid(s), point clouds"""
    src = os.path.join(ref, "testCases", case)
    keep = (".inp", ".xyz", ".dat")
    return sorted(f for f in os.listdir(src) if f.endswith(keep))


def parse_truth(text, case, procs):
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if f'SetRegressionCase("{case}")' in l)
    vecs, ignore, iterations = [], [], None
    for i in range(start, len(lines)):
        ln = lines[i]
        if "RunCase()" in ln:
            break
        m = re.search(r"SetNumberOfIterations\((\w+)\)", ln)
        if m:
            iterations = ITER_VARS[m.group(1)]
        m = re.search(r"SetIgnoreIndices\((\d+)\)", ln)
        if m:
            ignore.append(int(m.group(1)))
        if "SetResiduals" in ln:
            blob = ""
            for j in range(i, i + 6):
                blob += " " + lines[j]
                if "]" in lines[j]:
                    break
            nums = [float(v) for v in re.findall(r"[-+]?\d\.\d+e[-+]\d+", blob)]
            # the line the numbers start on
            first = next(j for j in range(i, i + 6) if re.search(r"\d\.\d+e[-+]\d+", lines[j]))
            vecs.append((nums, first + 1))
    # two vectors: `if Processors() == 2:` first, the 1-process one after `else:`
    truth, line = vecs[0] if (procs == 2 or len(vecs) == 1) else vecs[1]
    return dict(iterations=iterations, truth=truth, ignore=ignore, line=line)


def generate(ref, dst_root):
    text = open(os.path.join(ref, "testCases", "regressionTests.py")).read()
    truths = {}
    for case, procs in CASES.items():
        dst = os.path.join(dst_root, "cases", case)
        os.makedirs(dst, exist_ok=True)
        for f in data_files(ref, case):
            src = os.path.join(ref, "testCases", case, f)
            if case == "uniformFlow" and f.endswith(".inp"):
                deck = open(src, newline="").read()
                for pat, rep in UNIFORMFLOW_EDITS:
                    deck, n = re.subn(pat, rep, deck)
                    assert n >= 1, (pat, "no longer matches the reference's deck")
                with open(os.path.join(dst, f), "w", newline="") as fh:
                    fh.write(deck)
            else:
                shutil.copy(src, os.path.join(dst, f))
        if procs is not None:
            truths[case] = parse_truth(text, case, procs)
    return truths


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ref = args[0] if args else "/root/reference"
    write = "--write" in sys.argv
    committed = json.load(open(os.path.join(HERE, "regression_truths.json")))
    if write:
        truths = generate(ref, HERE)
        for case, t in truths.items():          # keep this repository's annotations
            committed.setdefault(case, {}).update(t)
        json.dump(committed, open(os.path.join(HERE, "regression_truths.json"), "w"), indent=1)
        print("wrote", ", ".join(CASES))
        return
    with tempfile.TemporaryDirectory() as tmp:
        truths = generate(ref, tmp)
        for case in CASES:
            a, b = os.path.join(tmp, "cases", case), os.path.join(HERE, "cases", case)
            names = sorted(os.listdir(a))
            assert names == sorted(os.listdir(b)), (case, names, sorted(os.listdir(b)))
            _, mismatch, errors = filecmp.cmpfiles(a, b, names, shallow=False)
            assert not mismatch and not errors, (case, mismatch, errors)
    assert sorted(os.listdir(os.path.join(HERE, "cases"))) == sorted(CASES)
    assert {k for k in committed if not k.startswith("_")} == set(truths)
    for case, t in truths.items():
        for key, val in t.items():
            assert committed[case][key] == val, (case, key, committed[case][key], val)
    print(f"tests/golden/ reproduced from {ref}: {len(CASES)} case directories "
          f"({sum(len(os.listdir(os.path.join(HERE, 'cases', c))) for c in CASES)} data files), "
          f"{len(truths)} truth vectors")


if __name__ == "__main__":
    main()
