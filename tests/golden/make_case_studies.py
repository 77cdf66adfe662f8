"""Fixture generator: the reference's two shipped example inputs ('case study 1', 'case study 2') as DATA.

Run here (the reference tree is present): python tests/golden/make_case_studies.py
Writes tests/golden/case_study_{1,2}.npz: coordinates (float64, as printed in atoms.xyz), species name per atom, the box line,
and the verbatim bytes of the three small parameter files (field.txt, control.txt, cuda.txt).  No reference source code is stored.
tests/util.py::materialise_case_study writes the four files back into a directory for aztot_init_md.
"""
import os

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for k in (1, 2):
        d = os.path.join(REF, "case study %d" % k)
        raw = open(os.path.join(d, "atoms.xyz"), newline="").read()
        crlf = "\r\n" in raw
        lines = raw.replace("\r\n", "\n").split("\n")
        n = int(lines[0])
        body = [ln.split() for ln in lines[2:2 + n]]
        names = sorted(set(b[0] for b in body))
        pos = np.array([[float(v) for v in b[1:4]] for b in body])
        out = os.path.join(HERE, "case_study_%d.npz" % k)
        np.savez_compressed(out, n=n, crlf=crlf, box_line=np.bytes_(lines[1]), names=np.array(names), name_idx=np.array([names.index(b[0]) for b in body], dtype=np.int8),
                            pos=pos, **{f.replace(".", "_"): np.bytes_(open(os.path.join(d, f), "rb").read()) for f in ("field.txt", "control.txt", "cuda.txt")})
        print(out, os.path.getsize(out))


if __name__ == "__main__":
    main()
