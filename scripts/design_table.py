"""Regenerate the measurement table of DESIGN.md section 5 from profiles/r02_summary.json (after scripts/make_profiles.py).
    python scripts/design_table.py          # rewrites the table in place"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r02_summary.json")))


def g(k):
    return d[k]


def M(k):
    return "%.1f M" % (g(k)["value_lookups_per_s"] / 1e6)


def ms(k, n=3):
    return ("%." + str(n) + "f") % g(k)["ms_per_step"]


def kus(k):
    return g(k)["kernel_ns"] / 1e3


def tb(k):
    return "%.2f" % (g(k)["GBps"] / 1e3)


def fr(k):
    return "%.3f" % g(k)["frac"]


def stb(k):
    return "%.2f" % (g(k)["step_GBps"] / 1e3)


def sfr(k):
    return "%.3f" % g(k)["step_frac"]


def pk(kk, name):
    return [v for k, v in kk.items() if name in k][0]


c5 = json.load(open(os.path.join(ROOT, "profiles", "r02_c5_shard.json")))["kernels"]
k5 = json.load(open(os.path.join(ROOT, "profiles", "r02_k5.json")))["kernels"]
lq = {(l["workload"], l["threshold"]): l for l in d["r02_long_queries"]["lines"]}
tr = d["r02_transpose"]["lines"]
kl = d["r02_k5"]["lines"][-1]
lpp = json.load(open(os.path.join(ROOT, "profiles", "r02_c3_exact.json")))["bench_same_run"]["roofline"]["launches_per_step"]
rows = f"""| workload (`profiles/…`) | lookups/s | ms/step | dominant kernel, rocprofv3 avg | achieved | frac of 8 TB/s |
|---|---|---|---|---|---|
| C3 exact, 8192 × 1 kbp (`r02_c3_exact`) | **{M('r02_c3_exact')}** | {ms('r02_c3_exact',2)} | `k_and_exact` {kus('r02_c3_exact')/1e3:.3f} ms × {lpp:.0f} | **{tb('r02_c3_exact')} TB/s** | **{fr('r02_c3_exact')}**; per step {sfr('r02_c3_exact')} |
| C3 threshold 0.4 (`r02_c3_t04`) | {M('r02_c3_t04')} | {ms('r02_c3_t04',2)} | `k_and_count<10,4>` {kus('r02_c3_t04')/1e3:.2f} ms | {tb('r02_c3_t04')} TB/s | {fr('r02_c3_t04')} |
| C3, 256 × 1 kbp per step (round 1's batch; `r02_c3_256x1kbp[_t04]`) | {M('r02_c3_256x1kbp')} / {M('r02_c3_256x1kbp_t04')} | {ms('r02_c3_256x1kbp')} / {ms('r02_c3_256x1kbp_t04')} | {kus('r02_c3_256x1kbp')/1e3:.3f} (per launch of 128 queries) / {kus('r02_c3_256x1kbp_t04')/1e3:.3f} ms | {tb('r02_c3_256x1kbp')} / {tb('r02_c3_256x1kbp_t04')} TB/s | {fr('r02_c3_256x1kbp')} / {fr('r02_c3_256x1kbp_t04')} |
| C2 1 M × 10 k, h=3, 1000 × 61-mers, a different batch every step (`r02_c2`): ONE launch per step, three in flight | **{M('r02_c2')}** under rocprofv3 (1300–1328 M unprofiled) | {ms('r02_c2',4)} (0.0234–0.0238) | `k_reads_fused<3,true>` {kus('r02_c2'):.1f} µs each — three overlapping, so a kernel's own duration spans its neighbours | {stb('r02_c2')} TB/s per step (`step_GBps`; {tb('r02_c2')} by the kernel's own clock) | **{sfr('r02_c2')}** per step (0.62–0.63 unprofiled; {fr('r02_c2')} by the kernel's own clock) |
| — the same with ONE read stream (`r02_c2_one_stream`, tuning build: no overlap, the kernel's own clock) | {M('r02_c2_one_stream')} | {ms('r02_c2_one_stream',4)} | `k_reads_fused<3,true>` {kus('r02_c2_one_stream'):.1f} µs alone on the device | {tb('r02_c2_one_stream')} TB/s | {fr('r02_c2_one_stream')} by the kernel's own clock |
| — the same through the three-launch route (`r02_c2_unfused`, tuning build) | {M('r02_c2_unfused')} | {ms('r02_c2_unfused',4)} | `k_and_exact` {kus('r02_c2_unfused'):.1f} µs (+ K1 {g('r02_c2_unfused')['k1_ms']*1e3:.1f}, K4 {g('r02_c2_unfused')['k4_ms']*1e3:.1f} µs with their event records) | {tb('r02_c2_unfused')} TB/s | {fr('r02_c2_unfused')} |
| C2 threshold 0.4 (`r02_c2_t04`) | {M('r02_c2_t04')} (1197–1217 M unprofiled) | {ms('r02_c2_t04',4)} (0.0255–0.0259) | `k_reads_fused<3,false>` {kus('r02_c2_t04'):.1f} µs, three overlapping | {stb('r02_c2_t04')} TB/s per step | {sfr('r02_c2_t04')} |
| C2 index, 32 768 reads per step, one launch (`r02_c2_32k_reads`) | {M('r02_c2_32k_reads')} | {ms('r02_c2_32k_reads')} | `k_reads_fused<3,true>` {kus('r02_c2_32k_reads')/1e3:.2f} ms, three overlapping | {stb('r02_c2_32k_reads')} TB/s per step | {sfr('r02_c2_32k_reads')} |
| C4 per-GPU shard 25 M × 62.5 k, h=3 (195 GB; `r02_c4_shard`) | {M('r02_c4_shard')} | {ms('r02_c4_shard')} | `k_and_exact` {kus('r02_c4_shard')/1e3:.3f} ms | {tb('r02_c4_shard')} TB/s | {fr('r02_c4_shard')} |
| C5 = that shard at threshold 0.4 **with score=True in the step**: hit lists to the host + K5 presence strings of 259 planted hits per batch, one batch behind the launches, K5 on a read stream beside the next batch's row-AND kernel (`r02_c5_shard`) | {M('r02_c5_shard')} | {ms('r02_c5_shard')} | `k_and_count<10,3>` {kus('r02_c5_shard')/1e3:.3f} ms (+ K5: bits {pk(c5,'presence_bits')['avg_ns']/1e3:.0f}, marks {pk(c5,'presence_pieces')['avg_ns']/1e3:.0f}, strings {pk(c5,'presence_expand<')['avg_ns']/1e3:.0f}, listed pieces {pk(c5,'expand_listed')['avg_ns']/1e3:.0f} µs while that kernel runs; alone 26 / 5 / 9 / 4 µs) | {tb('r02_c5_shard')} TB/s | {fr('r02_c5_shard')} |
| north-star per-GPU shard 10 M × 62.5 k, h=3 exact / 0.4 / h=4 (`r02_northstar_shard*`) | {M('r02_northstar_shard')} / {M('r02_northstar_shard_t04')} / {M('r02_northstar_shard_h4')} | {ms('r02_northstar_shard')} / {ms('r02_northstar_shard_t04')} / {ms('r02_northstar_shard_h4')} | {kus('r02_northstar_shard')/1e3:.3f} / {kus('r02_northstar_shard_t04')/1e3:.3f} / {kus('r02_northstar_shard_h4')/1e3:.3f} ms | {tb('r02_northstar_shard')} / {tb('r02_northstar_shard_t04')} / {tb('r02_northstar_shard_h4')} TB/s | {fr('r02_northstar_shard')} / {fr('r02_northstar_shard_t04')} / {fr('r02_northstar_shard_h4')} |
| one GPU's share of the default 8-GPU run: 10 M × 12.5 k, 8192 × 1 kbp (`r02_c3_strong8_shard`, with the 1-rank RCCL exchange `…_rccl1`) | {M('r02_c3_strong8_shard')} / {M('r02_c3_strong8_rccl1')} (against its shard) | {ms('r02_c3_strong8_shard',2)} / {ms('r02_c3_strong8_rccl1',2)} | `k_and_exact` {kus('r02_c3_strong8_shard')/1e3:.2f} / {kus('r02_c3_strong8_rccl1')/1e3:.2f} ms × 8; K1 {g('r02_c3_strong8_shard')['k1_ms']:.2f} ms | {tb('r02_c3_strong8_shard')} / {tb('r02_c3_strong8_rccl1')} TB/s | {fr('r02_c3_strong8_shard')} / {fr('r02_c3_strong8_rccl1')} |
| long queries on the C3 index, exact / 0.4 (`r02_long_queries`): 256 × 2 kbp (P=12) | {lq[('c3_q2000bp',1.0)]['lookups_per_s']/1e6:.1f} / {lq[('c3_q2000bp',0.4)]['lookups_per_s']/1e6:.1f} M | {lq[('c3_q2000bp',1.0)]['step_ms']:.2f} / {lq[('c3_q2000bp',0.4)]['step_ms']:.2f} | | {lq[('c3_q2000bp',1.0)]['GBps']/1e3:.2f} / {lq[('c3_q2000bp',0.4)]['GBps']/1e3:.2f} TB/s | {lq[('c3_q2000bp',1.0)]['frac']:.3f} / {lq[('c3_q2000bp',0.4)]['frac']:.3f} |
| — 128 × 4 kbp (P=12, pipelined counting loop) / 64 × 8 kbp (P=16, row lists sliced) | {lq[('c3_q4000bp',1.0)]['lookups_per_s']/1e6:.1f} / {lq[('c3_q4000bp',0.4)]['lookups_per_s']/1e6:.1f} M, {lq[('c3_q8000bp',1.0)]['lookups_per_s']/1e6:.1f} / {lq[('c3_q8000bp',0.4)]['lookups_per_s']/1e6:.1f} M | | | {lq[('c3_q4000bp',1.0)]['GBps']/1e3:.2f} / {lq[('c3_q4000bp',0.4)]['GBps']/1e3:.2f}, {lq[('c3_q8000bp',1.0)]['GBps']/1e3:.2f} / {lq[('c3_q8000bp',0.4)]['GBps']/1e3:.2f} TB/s | {lq[('c3_q4000bp',1.0)]['frac']:.3f} / {lq[('c3_q4000bp',0.4)]['frac']:.3f}, {lq[('c3_q8000bp',1.0)]['frac']:.3f} / {lq[('c3_q8000bp',0.4)]['frac']:.3f} |
| K5, 64 queries × 4080 hits (261 k strings of 970 characters; `r02_k5`) | — | {kl['kernels_ms']:.3f} (kernels) | `k_presence_bits` {pk(k5,'presence_bits')['max_ns']/1e6:.3f} ms + `k_presence_expand` {pk(k5,'presence_expand<')['max_ns']/1e6:.3f} ms + marks / listed pieces 0.009 ms | {kl['GBps']/1e3:.2f} TB/s (4.28–4.53 over boxes) | **{kl['frac']:.3f}** (0.535–0.566) |
| build transpose, filters resident: 10 M × 8192 / 1 M × 100 000 / 1 M × 99 963 (`r02_transpose`) | — | {tr[0]['kernels_ms']:.1f} / {tr[1]['kernels_ms']:.1f} / {tr[2]['kernels_ms']:.1f} | `k_transpose_tiles<2,2>` | {tr[0]['GBps']/1e3:.2f} / {tr[1]['GBps']/1e3:.2f} / {tr[2]['GBps']/1e3:.2f} TB/s in+out (3.3–4.1 over shapes and boxes: 16 384 columns 4.14) | {min(t['frac'] for t in tr):.2f}–{max(t['frac'] for t in tr):.2f} (0.42–0.52) |

"""
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index("| workload (`profiles/…`) | lookups/s | ms/step |")
b = s.index("* **PMC traffic**")
open(p, "w").write(s[:a] + rows + s[b:])
print(rows[:600])
