#!/usr/bin/env python3
"""Writes tests/golden/bunny_mesh.npz: the triangles the reference's reObjMesh builds from
Sandbox/res/bunny.obj_ group "bunny" (Sandbox/Sandbox.cpp:669-693 via Sandbox/WavefrontLoader.cpp:11-102):
for each face the first three vertices and the first three normals, as float64.
Container-only (reads /root/reference); the output is DATA (mesh vertices), not source."""
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Sandbox/res/bunny.obj_"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bunny_mesh.npz")


def face_vertex(tok):
    parts = tok.split("/")
    v = int(parts[0]) - 1
    vn = int(parts[2]) - 1 if len(parts) > 2 and parts[2] else -1
    return v, vn


def main():
    verts, norms, groups, cur = [], [], {}, "default"
    for line in open(REF):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            verts.append([float(x) for x in t[1:4]])
        elif t[0] == "vn":
            norms.append([float(x) for x in t[1:4]])
        elif t[0] == "g":
            cur = t[1]
        elif t[0] == "f":
            groups.setdefault(cur, []).append([face_vertex(x) for x in t[1:]])
    faces = groups["bunny"]
    V = np.zeros((len(faces), 9)); N = np.zeros((len(faces), 9))
    for i, f in enumerate(faces):
        for k in range(3):
            v, vn = f[k]
            V[i, k * 3:k * 3 + 3] = verts[v]
            N[i, k * 3:k * 3 + 3] = norms[vn] if vn >= 0 else 0.0
    np.savez_compressed(OUT, verts=V, normals=N)
    print(OUT, V.shape, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
