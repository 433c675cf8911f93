#!/usr/bin/env python3
"""Extract the shipped pixel-similarity parameters into plain JSON.

The reference keeps its trained Gaussian similarity parameters in
``parameters/pixel_similarity_clf/gauss.pkl`` (written by
``find_surfaces.py:210-213``, read by ``find_surfaces.py:221-226``).  The file is
a pickle of 12 ``numpy.float32`` scalars.  This script never unpickles it: it
walks the opcode stream with ``pickletools.genops`` (a disassembler that
executes nothing) and collects the twelve 4-byte payloads, which are the
little-endian float32 values in list order

    [w_1, w_2, w_3, bias, theta_1_1, theta_2_1, theta_3_1,
     theta_1_2, theta_2_2, theta_3_2, theta_2_3, theta_3_3]

The logistic-regression similarity parameters (``clf.pkl``, an sklearn object)
cannot be read without unpickling; its five numbers are taken from the opcode
stream the same way (one 32-byte and one 8-byte float64 buffer).

Run once in the build container:  python tools/extract_params.py
"""
import json
import os
import pickletools
import struct
import sys

REF = "/root/reference/parameters/pixel_similarity_clf"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "imagesegmentation_amd", "params")


def binbytes(path):
    data = open(path, "rb").read()
    out = []
    for op, arg, _pos in pickletools.genops(data):
        if op.name in ("SHORT_BINBYTES", "BINBYTES", "BINBYTES8"):
            out.append(bytes(arg))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    g = [b for b in binbytes(os.path.join(REF, "gauss.pkl")) if len(b) == 4]
    assert len(g) == 12, len(g)
    names = ["w_1", "w_2", "w_3", "bias", "theta_1_1", "theta_2_1", "theta_3_1",
             "theta_1_2", "theta_2_2", "theta_3_2", "theta_2_3", "theta_3_3"]
    doc = {
        "source": "parameters/pixel_similarity_clf/gauss.pkl (float32, little endian hex)",
        "order": names,
        "hex": [b.hex() for b in g],
        "value": [struct.unpack("<f", b)[0] for b in g],
    }
    with open(os.path.join(OUT, "gauss.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print("gauss:", doc["value"])

    bufs = binbytes(os.path.join(REF, "clf.pkl"))
    coef = [b for b in bufs if len(b) == 32]
    icpt = [b for b in bufs if len(b) == 8]
    if coef and icpt:
        doc = {
            "source": "parameters/pixel_similarity_clf/clf.pkl (sklearn LogisticRegression, float64 hex)",
            "coef_hex": coef[0].hex(),
            "intercept_hex": icpt[0].hex(),
            "coef": list(struct.unpack("<4d", coef[0])),
            "intercept": struct.unpack("<d", icpt[0])[0],
        }
        with open(os.path.join(OUT, "lr.json"), "w") as f:
            json.dump(doc, f, indent=1)
        print("lr:", doc["coef"], doc["intercept"])
    else:
        print("clf.pkl: float64 buffers not found", [len(b) for b in bufs], file=sys.stderr)


if __name__ == "__main__":
    main()
