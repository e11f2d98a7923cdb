"""Per-workload kernel measurements behind DESIGN.md section 5 / profiles/*.json (one JSON object per line on stdout).

    python scripts/measure.py [c2stream] [c3batch] [p16] [c4shard] [northstar] [k5] [transpose] ...

Every figure is a HIP-event duration recorded by the library around its own kernels (bigsi_hip_set_profiling) over
`reps` launches; run the same command under `rocprofv3 --kernel-trace --stats` for the per-kernel table that goes to
profiles/.  Algorithmic bytes follow SURVEY.md section 8d: unique rows x ceil(N/64) x 8 + result vectors."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bigsi_amd import _lib  # noqa: E402
from bigsi_amd._lib import check  # noqa: E402
from bigsi_amd.storage import get_storage  # noqa: E402

SEED = 20260928
PEAK = 8000.0


def open_index(name, m, n_cols, h, draws=2, k=31):
    st = get_storage({"storage-engine": "hip-hbm", "k": k, "m": m, "h": h,
                      "storage-config": {"name": name, "device": 0, "max_cols": n_cols}})
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    t0 = time.time()
    st.fill_synthetic(SEED, 0, draws)
    return st, time.time() - t0


def rand_seqs(rng, n, qlen):
    a = rng.integers(0, 4, size=(n, qlen), dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [lut[r].tobytes().decode("ascii") for r in a]


def stats(st, reset=1):
    s = _lib.Stats()
    check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(s), reset))
    return s


def alg_bytes(batch, n_seqs, n_cols):
    _, nu, _ = batch.unique()
    wv = -(-n_cols // 64)
    rows = 0
    for i in range(n_seqs):
        rows += np.unique(batch.rows(i, nu[i])).size
    return int(nu.sum()), rows, rows * wv * 8 + n_seqs * wv * 8


def run_steps(st, batches, threshold, reps, warm=3, prof=2, **kw):
    """(wall ms/step, Stats) over `reps` steps cycling through `batches`."""
    L = _lib.lib()
    for i in range(warm):
        batches[i % len(batches)].run(threshold, **kw)
    check(L.bigsi_hip_synchronize(st.handle))
    check(L.bigsi_hip_set_profiling(st.handle, prof))
    stats(st)
    t0 = time.perf_counter()
    for i in range(reps):
        batches[i % len(batches)].run(threshold, **kw)
    check(L.bigsi_hip_synchronize(st.handle))
    wall = (time.perf_counter() - t0) / reps * 1e3
    s = stats(st)
    check(L.bigsi_hip_set_profiling(st.handle, 0))
    return wall, s


def emit(name, **kw):
    kw = dict(workload=name, **kw)
    print(json.dumps(kw), flush=True)


def kernel_line(name, st, batches, n_cols, threshold, reps, note="", **kw):
    nseq = batches[0].n
    wall, s = run_steps(st, batches, threshold, reps, prof=2, **kw)            # row-AND kernel only: clean step time
    _, s1 = run_steps(st, batches, threshold, min(reps, 20), prof=1, **kw)     # every kernel group (event overhead in the step)
    lookups, rows, ab = alg_bytes(batches[0], nseq, n_cols)
    k2 = s.and_ms / reps                                    # row-AND time per step (large batches: several launches)
    r1 = min(reps, 20)
    emit(name, threshold=threshold, n_seqs=nseq, distinct_batches=len(batches), step_ms=wall, k2_ms=k2,
         k2_launches_per_step=s.and_launches / reps, k1_ms=s1.kmerize_ms / r1, k4_ms=s1.compact_ms / r1,
         lookups_per_batch=lookups, unique_rows=rows, alg_bytes=ab, GBps=ab / k2 / 1e6, frac=ab / k2 / 1e6 / PEAK,
         lookups_per_s=lookups / wall * 1e3, note=note)


def c2stream():
    """BASELINE configs[1] with a DIFFERENT batch every step: 32 staged batches of 1000 random 61-mers cycle, so the rows of
    a step (117 MB) are not the rows the previous steps left in the 256 MiB Infinity Cache."""
    m, n, h = 1_000_000, 10_000, 3
    st, _ = open_index("c2", m, n, h)
    rng = np.random.default_rng(7)
    same = [st.new_batch(rand_seqs(np.random.default_rng(1), 1000, 61), 31)]
    many = [st.new_batch(rand_seqs(rng, 1000, 61), 31) for _ in range(32)]
    for thr in (1.0, 0.4):
        kernel_line("c2_same_batch", st, same, n, thr, 200, note="one batch repeated: rows are cache resident")
        kernel_line("c2_stream", st, many, n, thr, 320, note="32 distinct batches cycling")
    for b in same + many:
        b.close()
    st.delete_all()


def c3batch():
    """C3 index; 256 vs 2048 vs 8192 queries per launch (does the address-ordered sweep survive a grid that is not co-resident?)."""
    m, n, h = 10_000_000, 100_000, 4
    st, _ = open_index("c3", m, n, h)
    rng = np.random.default_rng(1)
    for nq in (256, 2048, 8192):
        bs = [st.new_batch(rand_seqs(rng, nq, 1000), 31) for _ in range(2)]
        for thr in (1.0, 0.4):
            kernel_line("c3_batch%d" % nq, st, bs, n, thr, max(4, 5120 // nq), sparse_counts=True)
        for b in bs:
            b.close()
    st.delete_all()


def p16():
    """Counting kernel with P=16 planes (1024..65535 k-mers per query): 256 x 2 kbp and 128 x 4 kbp at t=0.4 vs exact."""
    m, n, h = 10_000_000, 100_000, 4
    st, _ = open_index("c3", m, n, h)
    rng = np.random.default_rng(3)
    for nq, ql in ((256, 1000), (256, 2000), (128, 4000), (64, 8000)):
        bs = [st.new_batch(rand_seqs(rng, nq, ql), 31) for _ in range(2)]
        for thr in (1.0, 0.4):
            kernel_line("c3_q%dbp" % ql, st, bs, n, thr, 12, sparse_counts=True)
        for b in bs:
            b.close()
    st.delete_all()


def shard(name, m, n, h):
    st, fill = open_index(name, m, n, h)
    rng = np.random.default_rng(1)
    bs = [st.new_batch(rand_seqs(rng, 256, 1000), 31) for _ in range(2)]
    for thr in (1.0, 0.4):
        kernel_line(name, st, bs, n, thr, 20, note="fill %.2f s" % fill, sparse_counts=True)
    for b in bs:
        b.close()
    st.delete_all()


def c4shard():
    shard("c4_shard_25Mx62500_h3", 25_000_000, 62_500, 3)


def northstar():
    shard("northstar_shard_10Mx62500_h3", 10_000_000, 62_500, 3)
    shard("northstar_shard_10Mx62500_h4", 10_000_000, 62_500, 4)


def k5():
    """Presence strings (score=True) at scale: 64 x 1 kbp queries on a 10M x 62.5k shard, each planted (70 % of its k-mers)
    into H samples, H = 16 .. 4096 per query -> up to 262k hits per batch; bigsi_hip_batch_presence_hits, kernel time from
    the library's events, whole-call time (host pair lists + H2D + kernels + D2H of the strings) from the wall clock."""
    m, n, h = 10_000_000, 62_500, 3
    st, _ = open_index("k5", m, n, h)
    rng = np.random.default_rng(11)
    seqs = rand_seqs(rng, 64, 1000)
    L = _lib.lib()
    done = 0
    for H in (16, 256, 4096):
        for qi, s in enumerate(seqs):
            for c in rng.choice(n, size=H - done, replace=False):
                st.insert_kmers(int(c), [s[:700]], 31)
        done = H
        b = st.new_batch(seqs, 31)
        b.run(0.4, sparse_counts=True)
        off, col, cnt = b.hits()
        nk, nu, _ = b.unique()
        b.presence_hits(off, col, nk)                      # warm (allocations)
        check(L.bigsi_hip_set_profiling(st.handle, 1))
        stats(st)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            blob, starts, lens = b.presence_hits(off, col, nk)
        dt = (time.perf_counter() - t0) / reps
        s_ = stats(st)
        check(L.bigsi_hip_set_profiling(st.handle, 0))
        kms = s_.presence_ms / max(s_.presence_launches, 1)
        ab = s_.presence_bytes / max(s_.presence_launches, 1)
        for t in range(8):
            assert int((blob[int(starts[t]):int(starts[t] + lens[t])] == ord("1")).sum()) >= 670
        emit("k5_presence_hits", n_seqs=len(seqs), hits=int(off[-1]), positions=int(nk[0]), kernels_ms=kms, call_ms=dt * 1e3,
             alg_bytes=ab, GBps=ab / kms / 1e6, frac=ab / kms / 1e6 / PEAK, string_bytes=int(lens.sum()),
             note="k_presence_bits + k_presence_expand; alg bytes = unique k-mers x h x 8 x distinct hit words + string bytes")
        # K6: the same hits through bigsi_hip_batch_score_hits -- packed presence bits + score records instead of ASCII strings
        b.score_hits(off, col, cnt, nk)
        check(L.bigsi_hip_set_profiling(st.handle, 1))
        stats(st)
        t0 = time.perf_counter()
        for _ in range(reps):
            rec, pbits, boff = b.score_hits(off, col, cnt, nk)
        dt = (time.perf_counter() - t0) / reps
        s_ = stats(st)
        check(L.bigsi_hip_set_profiling(st.handle, 0))
        kms = s_.presence_ms / max(s_.presence_launches, 1)
        ab = s_.presence_bytes / max(s_.presence_launches, 1)
        from bigsi_amd.graph.bigsi import scored_rows
        t0 = time.perf_counter()
        rows = scored_rows(rec, pbits, boff, np.repeat(nk.astype(np.int64), np.diff(off.astype(np.int64))), n)
        host_s = time.perf_counter() - t0
        assert len(rows) == int(off[-1]) and rows[0][2].count("1") >= 670
        emit("k6_score_hits", n_seqs=len(seqs), hits=int(off[-1]), positions=int(nk[0]), kernels_ms=kms, call_ms=dt * 1e3,
             alg_bytes=ab, GBps=ab / kms / 1e6, frac=ab / kms / 1e6 / PEAK, presence_bits_bytes=int(boff[-1]), score_bytes=int(rec.nbytes),
             host_rows_ms=host_s * 1e3, host_us_per_hit=host_s / max(int(off[-1]), 1) * 1e6,
             note="k_presence_bits + k_presence_score; alg bytes = unique k-mers x h x 8 x distinct hit words + packed bits + score records; "
                  "host_rows = closed-form fields + presence strings for all hits (scored_rows)")
        b.close()
    st.delete_all()


def transpose():
    """Bloom filters -> matrix columns (the build transpose): filters resident in device memory
    (bigsi_hip_insert_columns_device), kernel time from the library's events; bytes = filters in + rows out."""
    import torch
    L = _lib.lib()
    shapes = ((10_000_000, 8192), (1_000_000, 100_000), (1_000_000, 100_000 - 37))
    if os.environ.get("BIGSI_TR_SHAPES"):          # e.g. "4000000x16384,2000000x32768"
        shapes = tuple(tuple(int(x) for x in sh.split("x")) for sh in os.environ["BIGSI_TR_SHAPES"].split(","))
    for m, ncols in shapes:
        st = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": m, "h": 3,
                          "storage-config": {"name": "tr", "device": 0, "max_cols": ncols}})
        st.delete_all()
        for key, v in (("number_of_rows", m), ("number_of_cols", 0), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", 3)):
            st.set_integer(key, v)
        nb = (m + 7) // 8
        align = int(os.environ.get("BIGSI_TR_PITCH_ALIGN", "128"))      # the library stages host filters at a 128-byte pitch (whole lines per run)
        pitch = -(-nb // align) * align
        blooms = torch.randint(0, 256, (ncols, pitch), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        check(L.bigsi_hip_insert_columns_device(st.handle, 0, 512, blooms.data_ptr(), pitch))      # warm
        check(L.bigsi_hip_set_profiling(st.handle, 1))
        stats(st)
        check(L.bigsi_hip_insert_columns_device(st.handle, 0, ncols, blooms.data_ptr(), pitch))
        s = stats(st)
        check(L.bigsi_hip_set_profiling(st.handle, 0))
        moved = 2 * ncols * nb
        st.res.written[:] = True
        for r in (() if os.environ.get("BIGSI_HIP_TR_SKIP") else (0, 1, 511, 512, m // 2 + 3, m - 1)):
            bits = ((blooms[:, r >> 3] >> (7 - (r & 7))) & 1).cpu().numpy()
            assert np.array_equal(st.get_rows_packed([r], (ncols + 7) // 8)[0], np.packbits(bits)), r
        emit("transpose_device", m=m, cols=ncols, kernels_ms=s.transpose_ms, bytes_in_plus_out=moved,
             GBps=moved / s.transpose_ms / 1e6, frac=moved / s.transpose_ms / 1e6 / PEAK, filter_pitch=pitch,
             note="k_transpose_regs (+ k_insert_columns for ragged edges); filters resident in HBM")
        del blooms
        st.delete_all()


if __name__ == "__main__":
    todo = sys.argv[1:] or ["c2stream", "c3batch", "p16", "c4shard", "northstar", "k5", "transpose"]
    for t in todo:
        globals()[t]()
