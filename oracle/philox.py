"""Philox4x32-10 counter RNG and the uniform construction of the RBM sampler contract.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package (``keras_unsupervised_amd``); only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg use it, as the checker.

Why it exists: the reference draws its Bernoulli thresholds with
``K.random_uniform`` (``ku/ebm/rbm.py:46,52,82,121``), i.e. TensorFlow's Philox
stream, which is a third-party dependency absent from ``/root/reference`` and not
observable here.  The build therefore fixes its *own* counter
contract, stated once here and mirrored by the HIP kernels:

    block   = philox4x32_10(key=(seed_lo, seed_hi),
                            ctr=(col, row >> 2, stream_id, step))
    u[row, col] = u32_to_unit_float(block[row & 3])
    u32_to_unit_float(x) = bitcast_f32((x & 0x7FFFFF) | 0x3F800000) - 1.0f   in [0, 1)

``row`` is the *global* row index of the batch (so 1-GPU and N-GPU runs draw the
same numbers), ``col`` the unit index, ``stream_id`` names the sampling site of
the Gibbs chain and ``step`` the parameter-update (or call) counter.  The four
words of one block feed four consecutive rows of one column, which is exactly
the 4 accumulator registers a lane owns in every gfx950 MFMA C/D layout.

Known-answer vectors (Random123 ``kat_vectors``; also listed in SURVEY.md 8(c))
are checked in ``tests/test_oracle_philox.py``.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)
_SH32 = np.uint64(32)


def philox4x32_10(ctr, key):
    """Vectorised Philox4x32-10.

    ctr: tuple of four array-likes (broadcastable) of uint32 values
    key: tuple (k0, k1) of python ints / uint32 scalars
    returns: list of four uint32 arrays (broadcast shape)
    """
    c0, c1, c2, c3 = np.broadcast_arrays(*[np.asarray(c, dtype=np.uint64) & _MASK32 for c in ctr])
    k0 = int(key[0]) & 0xFFFFFFFF
    k1 = int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> _SH32, p0 & _MASK32
        hi1, lo1 = p1 >> _SH32, p1 & _MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def u32_to_unit_float(x):
    """Mantissa-fill construction of a float32 in [0,1) from 23 low bits."""
    x = np.asarray(x, dtype=np.uint32)
    bits = (x & np.uint32(0x7FFFFF)) | np.uint32(0x3F800000)
    return bits.view(np.float32) - np.float32(1.0)


def block_words(rows, cols, seed, stream_id, step, row0=0):
    """Raw uint32 Philox word for every element of a [rows, cols] matrix.

    One block per (column, group of 4 rows); word w of the block is row 4*group + w.
    """
    g_lo, g_hi = row0 >> 2, (row0 + rows + 3) >> 2
    rg = np.arange(g_lo, g_hi, dtype=np.uint64)[:, None]
    c = np.arange(cols, dtype=np.uint64)[None, :]
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    w = philox4x32_10((c, rg & _MASK32, np.uint64(stream_id & 0xFFFFFFFF), np.uint64(step & 0xFFFFFFFF)), key)
    full = np.stack(w, axis=1).reshape(4 * (g_hi - g_lo), cols)   # [group, word, col] -> rows
    off = row0 - 4 * g_lo
    return np.ascontiguousarray(full[off:off + rows])


def uniform(rows, cols, seed, stream_id, step, row0=0):
    """float32 uniforms in [0,1) for a [rows, cols] matrix under the contract above."""
    return u32_to_unit_float(block_words(rows, cols, seed, stream_id, step, row0))


def normal(rows, cols, seed, stream_id, step, row0=0):
    """Standard normals by Box-Muller from two uniform planes of the same site.

    Plane A is ``stream_id``, plane B is ``stream_id | 0x80000000``:
        z = sqrt(-2 ln(1 - uA)) * cos(2 pi uB)
    (1 - uA lies in (0, 1], so the log is finite.)  Evaluated in float32, the
    arithmetic the kernels use.
    """
    ua = uniform(rows, cols, seed, stream_id, step, row0)
    ub = uniform(rows, cols, seed, (stream_id | 0x80000000) & 0xFFFFFFFF, step, row0)
    rad = np.sqrt(np.float32(-2.0) * np.log(np.float32(1.0) - ua, dtype=np.float32), dtype=np.float32)
    return (rad * np.cos(np.float32(2.0 * np.pi) * ub, dtype=np.float32)).astype(np.float32)
