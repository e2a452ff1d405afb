"""Seeded structural stand-ins for the SuiteSparse matrices BASELINE.json config 3 names (the .mtx files
are neither in the reference tree nor reachable from the GPU box: SURVEY.md section 8d).  numpy only,
vectorised; every generator returns (m, row_ptr, col_ind, val) with 0-based sorted, duplicate-free rows
and a full diagonal.  Results obtained on them are labelled "stand-in" wherever they are reported."""
import numpy as np


def _to_csr(n, rows, cols, seed):
    key = np.unique(rows.astype(np.int64) * n + cols.astype(np.int64))
    r = (key // n).astype(np.int64)
    c = (key % n).astype(np.int32)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(row_ptr, r + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    rng = np.random.default_rng(seed)
    val = rng.uniform(-1.0, 1.0, size=len(c))
    val[c == r] = rng.uniform(4.0, 8.0, size=int(np.sum(c == r)))
    return n, row_ptr.astype(np.int32), c, val


def _powerlaw_lengths(rng, n, mean, maxlen, alpha):
    ln = (rng.pareto(alpha, size=n) + 1.0)
    ln = ln * (mean / ln.mean())
    return np.clip(np.rint(ln), 1, maxlen).astype(np.int64)


def circuit_like(n=170998, seed=101):
    """scircuit-like: n=170,998, ~5.6 nnz/row, power-law rows (max ~353), 20 % near-diagonal."""
    rng = np.random.default_rng(seed)
    ln = _powerlaw_lengths(rng, n, 4.7, 353, 1.6)
    ln[rng.integers(0, n, 3)] = 353
    rows = np.repeat(np.arange(n), ln)
    near = rng.random(len(rows)) < 0.2
    cols = rng.integers(0, n, len(rows))
    cols[near] = np.clip(rows[near] + rng.integers(-30, 31, int(near.sum())), 0, n - 1)
    rows = np.concatenate([rows, np.arange(n)])
    cols = np.concatenate([cols, np.arange(n)])
    return _to_csr(n, rows, cols, seed + 1)


def web_like(n=1000005, seed=202):
    """webbase-1M-like: n=1,000,005, ~3.1 nnz/row, a few rows of ~4,700, Zipf-distributed columns."""
    rng = np.random.default_rng(seed)
    ln = _powerlaw_lengths(rng, n, 2.2, 4700, 1.3)
    ln[rng.integers(0, n, 4)] = 4700
    rows = np.repeat(np.arange(n), ln)
    cols = (rng.zipf(1.3, len(rows)) - 1) % n
    mix = rng.random(len(rows)) < 0.5
    cols[mix] = rng.integers(0, n, int(mix.sum()))
    rows = np.concatenate([rows, np.arange(n)])
    cols = np.concatenate([cols, np.arange(n)])
    return _to_csr(n, rows, cols, seed + 1)


def _block_csr(nb, bs, bcol, ok, seed):
    """CSR of a block matrix with dense bs x bs blocks, built directly in sorted order: block row i couples to the block
    columns bcol[i, k] (ascending in k) where ok[i, k]; scalar row bs*i + a holds bs*bcol + b for b = 0..bs-1.
    Values as _to_csr draws them (same generator calls in the same order, so the matrices are those of the original
    sort-based construction, bit for bit)."""
    n = nb * bs
    cnt = ok.sum(axis=1).astype(np.int64) * bs  # entries per scalar row of block row i
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum(np.repeat(cnt, bs))
    sel = np.broadcast_to(ok[:, None, :, None], (nb, bs, ok.shape[1], bs))
    cols = (bcol.astype(np.int32)[:, None, :, None] * np.int32(bs) + np.arange(bs, dtype=np.int32)[None, None, None, :])
    cols = np.broadcast_to(cols, sel.shape)[sel]
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(row_ptr))
    rng = np.random.default_rng(seed)
    val = rng.uniform(-1.0, 1.0, size=len(cols))
    diag = cols == rows
    val[diag] = rng.uniform(4.0, 8.0, size=int(diag.sum()))
    return n, row_ptr.astype(np.int32), cols, val


def shell_like(n=1508065, seed=303, width=600):
    """af_shell10-like: n=1,508,065, 5 dofs per node of a structured shell mesh `width` nodes wide; node
    (i, j) couples to (i, j+-1), (i+-1, j) and the (i-1, j-1) / (i+1, j+1) diagonal: 7 dense 5x5 blocks,
    ~35 nnz/row, very uniform.  No coupling across the ends of a mesh row."""
    nb = n // 5
    bi = np.arange(nb, dtype=np.int64)
    j = bi % width
    offs = np.array([-width - 1, -width, -1, 0, 1, width, width + 1], dtype=np.int64)
    dj = np.array([-1, 0, -1, 0, 1, 0, 1], dtype=np.int64)
    bc = bi[:, None] + offs[None, :]
    jj = j[:, None] + dj[None, :]
    ok = (bc >= 0) & (bc < nb) & (jj >= 0) & (jj < width)
    return _block_csr(nb, 5, bc, ok, seed)


def flan_like(nx=81, ny=80, nz=80, seed=404):
    """Flan_1565-like: 3-D 27-point stencil with 3 dofs per node, n = 3*nx*ny*nz = 1,555,200, ~75/row."""
    nodes = nx * ny * nz
    idx = np.arange(nodes, dtype=np.int64)
    ix, iy, iz = idx // (ny * nz), (idx // nz) % ny, idx % nz
    bc = np.empty((nodes, 27), dtype=np.int64)
    ok = np.empty((nodes, 27), dtype=bool)
    k = 0
    for dx in (-1, 0, 1):  # ascending neighbour index: dx, then dy, then dz
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                ok[:, k] = ((ix + dx >= 0) & (ix + dx < nx) & (iy + dy >= 0) & (iy + dy < ny)
                            & (iz + dz >= 0) & (iz + dz < nz))
                bc[:, k] = (ix + dx) * (ny * nz) + (iy + dy) * nz + (iz + dz)
                k += 1
    return _block_csr(nodes, 3, bc, ok, seed)


# ---- round 3: the same four, OFF their ideal ordering ------------------------------------------------
# The mesh stand-ins above are perfectly structured meshes in natural node order -- the best case for shared / shifted column
# lists and for the supernodal TRSV; the graph stand-ins draw most columns uniformly at random -- harsher than a real circuit
# or web graph, which have locality.  The variants below move each pair towards what a real SuiteSparse ordering looks like,
# so that the mix / TRSV legs show how the format tricks degrade (mesh) and what locality buys (graphs).
def _unstructure(bc, ok, seed, window=256, drop_permille=100):
    """irregular valence + a random renumbering of the nodes inside windows of `window` consecutive nodes.
    bc / ok: neighbour block ids (nb x K, any order) and their validity.  A coupling {i, j} is dropped (both directions, never
    the diagonal) when a symmetric hash of the pair falls below drop_permille / 1000; then node i becomes perm[i], a
    permutation that shuffles every window.  Returns (bc, ok) with each row's valid neighbours ascending and first."""
    nb, K = bc.shape
    i = np.arange(nb, dtype=np.int64)[:, None]
    lo, hi = np.minimum(i, bc), np.maximum(i, bc)
    h = (lo * 2654435761 + hi * 40503 + seed) % 1000
    ok = ok & ((h >= drop_permille) | (bc == i))
    rng = np.random.default_rng(seed)
    perm = np.arange(nb, dtype=np.int64)
    for w0 in range(0, nb, window):
        w1 = min(nb, w0 + window)
        perm[w0:w1] = w0 + rng.permutation(w1 - w0)
    inv = np.argsort(perm)
    bc2 = perm[np.clip(bc, 0, nb - 1)][inv]
    ok2 = ok[inv]
    bc2 = np.where(ok2, bc2, nb + np.arange(K, dtype=np.int64)[None, :])  # invalid ones last, distinct
    order = np.argsort(bc2, axis=1, kind="stable")
    return np.take_along_axis(bc2, order, 1), np.take_along_axis(ok2, order, 1)


def shell_like_unstructured(n=1508065, seed=313, width=600):
    """shell-like with ~10 % of the node couplings removed and the nodes renumbered at random inside windows of 256: rows of
    a node still share one column list (5 dofs), but lists no longer repeat from node to node and valence varies"""
    nb = n // 5
    bi = np.arange(nb, dtype=np.int64)
    j = bi % width
    offs = np.array([-width - 1, -width, -1, 0, 1, width, width + 1], dtype=np.int64)
    dj = np.array([-1, 0, -1, 0, 1, 0, 1], dtype=np.int64)
    bc = bi[:, None] + offs[None, :]
    ok = (bc >= 0) & (bc < nb) & ((j[:, None] + dj[None, :]) >= 0) & ((j[:, None] + dj[None, :]) < width)
    bc, ok = _unstructure(bc, ok, seed)
    return _block_csr(nb, 5, bc, ok, seed)


def flan_like_unstructured(nx=81, ny=80, nz=80, seed=414):
    """flan-like (27-point, 3 dofs) with ~10 % of the couplings removed and windowed random renumbering"""
    nodes = nx * ny * nz
    idx = np.arange(nodes, dtype=np.int64)
    ix, iy, iz = idx // (ny * nz), (idx // nz) % ny, idx % nz
    bc = np.empty((nodes, 27), dtype=np.int64)
    ok = np.empty((nodes, 27), dtype=bool)
    k = 0
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                ok[:, k] = ((ix + dx >= 0) & (ix + dx < nx) & (iy + dy >= 0) & (iy + dy < ny)
                            & (iz + dz >= 0) & (iz + dz < nz))
                bc[:, k] = (ix + dx) * (ny * nz) + (iy + dy) * nz + (iz + dz)
                k += 1
    bc, ok = _unstructure(bc, ok, seed)
    return _block_csr(nodes, 3, bc, ok, seed)


def circuit_like_local(n=170998, seed=111):
    """circuit-like with locality: the same power-law row lengths, 80 % of the columns within +-2,000 of the row (a circuit
    netlist in a reasonable ordering), 20 % anywhere (global nets)"""
    rng = np.random.default_rng(seed)
    ln = _powerlaw_lengths(rng, n, 4.7, 353, 1.6)
    ln[rng.integers(0, n, 3)] = 353
    rows = np.repeat(np.arange(n), ln)
    cols = rng.integers(0, n, len(rows))
    near = rng.random(len(rows)) < 0.8
    cols[near] = np.clip(rows[near] + rng.integers(-2000, 2001, int(near.sum())), 0, n - 1)
    rows = np.concatenate([rows, np.arange(n)])
    cols = np.concatenate([cols, np.arange(n)])
    return _to_csr(n, rows, cols, seed + 1)


def web_like_local(n=1000005, seed=212):
    """web-like with locality: 70 % of the links inside the page's own "site" (a window of +-5,000 rows: a crawl orders pages
    host by host), 20 % to Zipf-distributed hubs, 10 % anywhere; same row-length law and the same few rows of ~4,700"""
    rng = np.random.default_rng(seed)
    ln = _powerlaw_lengths(rng, n, 2.2, 4700, 1.3)
    ln[rng.integers(0, n, 4)] = 4700
    rows = np.repeat(np.arange(n), ln)
    u = rng.random(len(rows))
    cols = (rng.zipf(1.3, len(rows)) - 1) % n
    far = u < 0.1
    cols[far] = rng.integers(0, n, int(far.sum()))
    near = u >= 0.3
    cols[near] = np.clip(rows[near] + rng.integers(-5000, 5001, int(near.sum())), 0, n - 1)
    rows = np.concatenate([rows, np.arange(n)])
    cols = np.concatenate([cols, np.arange(n)])
    return _to_csr(n, rows, cols, seed + 1)


# ---- round 4: a BLOCK-DENSE matrix --------------------------------------------------------------------
# The SuiteSparse four have 1 / 3 / 5 unknowns per node: cut into the 16 x 16 tiles of v_mfma_f64_16x16x4_f64 their fill is
# 0.19-0.31 (profiles/r2/mfma_f64_probe.jsonl).  What the north star's "blocked-ELL variant that feeds MFMA only where nnz/row
# is uniform enough to form dense tiles" is for is a multi-physics / high-order discretisation with many unknowns per node:
# here 16 per node on a 3-D grid of nodes with the 7-point node stencil => 7 dense 16 x 16 blocks per block row, 112 entries
# per row.  `keep` < 1 thins every block at random (the diagonal stays) to that fill.
def block_dense(nx=32, ny=32, nz=32, dofs=16, keep=1.0, seed=505):
    nodes = nx * ny * nz
    idx = np.arange(nodes, dtype=np.int64)
    ix, iy, iz = idx // (ny * nz), (idx // nz) % ny, idx % nz
    offs = [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 0), (0, 0, 1), (0, 1, 0), (1, 0, 0)]  # ascending node index
    bc = np.empty((nodes, 7), dtype=np.int64)
    ok = np.empty((nodes, 7), dtype=bool)
    for k, (dx, dy, dz) in enumerate(offs):
        ok[:, k] = ((ix + dx >= 0) & (ix + dx < nx) & (iy + dy >= 0) & (iy + dy < ny) & (iz + dz >= 0) & (iz + dz < nz))
        bc[:, k] = (ix + dx) * (ny * nz) + (iy + dy) * nz + (iz + dz)
    n, rp, ci, v = _block_csr(nodes, dofs, bc, ok, seed)
    if keep >= 1.0:
        return n, rp, ci, v
    rng = np.random.default_rng(seed + 1)
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
    sel = (rng.random(len(ci)) < keep) | (ci == rows)
    rp2 = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rp2, rows[sel].astype(np.int64) + 1, 1)
    return n, np.cumsum(rp2).astype(np.int32), ci[sel], v[sel]


def block_dense_75():
    return block_dense(keep=0.75, seed=515)


ALL = {"block-dense": block_dense, "block-dense, 75 % fill": block_dense_75,
       "circuit-like": circuit_like, "web-like": web_like, "shell-like": shell_like, "flan-like": flan_like,
       "circuit-like, local": circuit_like_local, "web-like, local": web_like_local,
       "shell-like, unstructured": shell_like_unstructured, "flan-like, unstructured": flan_like_unstructured}


# ---- MatrixMarket input (tests/include/aoclsparse_init.hpp:452-694 reads coordinate real / integer / pattern,
# general or symmetric expanded to full) -------------------------------------------------------------------
REAL_FILES = {"circuit-like": "scircuit.mtx", "web-like": "webbase-1M.mtx", "shell-like": "af_shell10.mtx",
              "flan-like": "Flan_1565.mtx"}


def read_mtx(path, seed=7):
    """MatrixMarket coordinate file -> (m, n, row_ptr, col_ind, val), 0-based CSR with sorted rows.  Symmetric /
    skew-symmetric / hermitian storage is expanded to the full matrix; pattern files get seeded random values;
    duplicate entries are summed (as a coordinate assembly would)."""
    with open(path, "rb") as f:
        header = f.readline().decode().lower().split()
        if len(header) < 5 or header[0] != "%%matrixmarket" or header[1] != "matrix" or header[2] != "coordinate":
            raise ValueError("%s: only 'matrix coordinate' MatrixMarket files are supported" % path)
        field, symm = header[3], header[4]
        if field == "complex":
            raise ValueError("%s: complex matrices are out of scope" % path)
        line = f.readline()
        while line.startswith(b"%") or not line.strip():
            line = f.readline()
        m, n, nnz = (int(t) for t in line.split()[:3])
        data = np.loadtxt(f, ndmin=2) if nnz else np.zeros((0, 3))
    r, c = data[:, 0].astype(np.int64) - 1, data[:, 1].astype(np.int64) - 1
    v = np.random.default_rng(seed).uniform(-1, 1, len(r)) if field == "pattern" else data[:, 2].astype(np.float64)
    if symm in ("symmetric", "hermitian", "skew-symmetric"):
        off = r != c
        r, c = np.concatenate([r, c[off]]), np.concatenate([c, r[off]])
        v = np.concatenate([v, -v[off] if symm == "skew-symmetric" else v[off]])
    key = r * n + c
    order = np.argsort(key, kind="stable")
    key, v = key[order], v[order]
    uniq, start = np.unique(key, return_index=True)
    v = np.add.reduceat(v, start) if len(v) else v
    rows = (uniq // n).astype(np.int64)
    row_ptr = np.zeros(m + 1, dtype=np.int64)
    np.add.at(row_ptr, rows + 1, 1)
    return m, n, np.cumsum(row_ptr).astype(np.int32), (uniq % n).astype(np.int32), v


def load(name):
    """(label, m, row_ptr, col_ind, val): the real SuiteSparse matrix when $MATRIX_DIR holds its .mtx file, else the
    seeded stand-in (labelled as such)."""
    import os
    d = os.environ.get("MATRIX_DIR")
    if d and name in REAL_FILES and os.path.isfile(os.path.join(d, REAL_FILES[name])):
        m, n, rp, ci, v = read_mtx(os.path.join(d, REAL_FILES[name]))
        assert m == n
        return REAL_FILES[name], m, rp, ci, v
    m, rp, ci, v = ALL[name]()
    return name + " (stand-in)", m, rp, ci, v
