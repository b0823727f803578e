"""TEST INFRASTRUCTURE -- sequential CPU statements of the mesh-side helpers of the path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

patch_grow: /root/reference/code/data_util.py:55-84 (mesh_get_neighbor_np), restated in C (oracle/oracle_c.c:
oracle_patch_grow) and PINNED by the face lists the reference's own function produced
(tests/golden/patches_n8.npz; tests/test_oracle_golden.py).  split_faces: the seed loop around it,
/root/reference/code/dataset.py:156-193.
"""
import ctypes
import os

import numpy as np

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_build', 'liboracle_c.so')
        _LIB = ctypes.CDLL(path)
    return _LIB


def patch_grow(fv, vf_rowptr, vf_list, seed, neighbor_count=None, ring_count=None):
    """fv [F,3], vf_rowptr [V+1], vf_list int32 host arrays -> face ids in the reference's visiting order."""
    fv = np.ascontiguousarray(fv, dtype=np.int32)
    rp = np.ascontiguousarray(vf_rowptr, dtype=np.int32)
    ls = np.ascontiguousarray(vf_list, dtype=np.int32)
    F = fv.shape[0]
    out = np.empty(F, dtype=np.int32)
    sel = np.empty(F, dtype=np.uint8)
    n = ctypes.c_int64(0)
    _lib().oracle_patch_grow(ctypes.c_void_p(fv.ctypes.data), ctypes.c_void_p(rp.ctypes.data), ctypes.c_void_p(ls.ctypes.data),
                             ctypes.c_int64(F), ctypes.c_int64(int(seed)), ctypes.c_int64(int(neighbor_count or 0)),
                             ctypes.c_int64(int(ring_count or 0)), ctypes.c_void_p(sel.ctypes.data),
                             ctypes.c_void_p(out.ctypes.data), ctypes.byref(n))
    return out[:n.value].copy()


def split_faces(d2, fv, vf_rowptr, vf_list, submesh_size):
    """dataset.py:156-193: list of (seed, face ids) -- seed = unvisited face with the largest d2 (np.argmax: first of ties)."""
    F = fv.shape[0]
    flag = np.zeros(F, dtype=bool)
    seed = int(np.argmax(d2))
    out = []
    while True:
        sel = patch_grow(fv, vf_rowptr, vf_list, seed, neighbor_count=submesh_size)
        flag[sel] = True
        out.append((seed, sel))
        left = np.where(~flag)[0]
        if left.size == 0:
            return out
        seed = int(left[np.argmax(d2[left])])
