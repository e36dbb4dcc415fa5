#!/usr/bin/env python
"""profiles/pmc.json from the rocprofv3 --pmc passes written by tools/rocprof_passes.sh (separate passes, no tracing).

Per workload and hot kernel (and per variant where the kernel name tells them apart):
  traffic_bytes   (2 * FETCH_SIZE + WRITE_SIZE) KiB per launch: on gfx950 FETCH_SIZE reports half the bytes of a wide
                  (16 B/lane) coalesced read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  Fabric-side: hits in
                  the 256 MB Infinity Cache are counted too.
  mfma_util       SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * kernel cycles): the share of matrix-pipe cycles in use.  The
                  busy counter is summed over all SIMDs (it equals SQ_INSTS_MFMA x 32 cycles for v_mfma_f32_32x32x16_bf16);
                  kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs).
  valu_util       4 * SQ_ACTIVE_INST_VALU / (1024 * kernel cycles)  (the SQ_ACTIVE_* / SQ_WAVE_CYCLES counters tick once per
                  4 cycles: SQ_WAVE_CYCLES * 4 / waves reproduces the in-kernel s_memtime stamps)
  wait_frac       SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of wave time parked at s_waitcnt / s_barrier
  l2_hit          TCC_HIT / (TCC_HIT + TCC_MISS)
    python tools/make_pmc_json.py gpurun_out/prof_<tag> <workload> [profiles/pmc.json]
Also (re)writes traffic.json next to the output: the `traffic_bytes` column of pmc.json alone, per workload and kernel[:variant] -
ONE pass set feeds both files (tests/test_profiles.py checks that they agree).
"""
import csv, glob, json, os, sys
from collections import defaultdict

N_SIMD, N_XCD = 1024, 8


def name(n):
    if 'gat::fwd' in n: return 'forward'                    # fused GAT_plus layer kernels (csrc/gadapt_gat.inc)
    if 'gat::bwd_t' in n: return 'backward_target'
    if 'gat::bwd_s' in n: return 'backward_source'
    if 'grand_fwd' in n or 'wide::fwd' in n: return 'forward'
    if "bwd_target" in n: return "backward_target"          # grand_bwd_target_kernel<...>, grand_bwd_target_compact_kernel<SUMS>
    if 'bwd_source' in n: return 'backward_source'          # grand_bwd_source_kernel<...>, grand_bwd_source4_kernel<C, GC>
    return None


def variant(n):
    """Variant names of bench.py (VARIANT_NAMES) from the template arguments in the kernel name: 'dense' where the name says so,
    None where the name cannot tell (the tiled forward's head-only output is a run-time argument): callers skip those."""
    try:
        a = n[n.index('<') + 1:n.index('>')].replace(' ', '').split(',')
    except ValueError:
        return None
    if 'bwd_target_compact' in n:
        return 'compact_x'
    if 'bwd_target' in n and len(a) <= 4:                   # round 6 on: <C, SUMS, GC, D4>
        a += ['false'] * (4 - len(a))
        if a[3] == 'true': return 'compact_g+out4' if a[2] == 'true' else 'out4'
        return 'compact_g' if a[2] == 'true' else 'dense'
    if 'bwd_target' in n:                                   # rounds 1-5: <C, SUMS, GC, XC, DA, D4[, STR]>
        a += ['false'] * (6 - len(a))
        if a[3] == 'true': return 'compact_x'
        if a[5] == 'true': return 'compact_g+out4' if a[2] == 'true' else 'out4'
        return 'compact_g' if a[2] == 'true' else 'dense'
    if 'bwd_source4' in n:                                  # <C, GC>
        return 'compact_g+out4' if len(a) > 1 and a[1] == 'true' else 'out4'
    if 'bwd_source' in n:
        return 'compact_g' if len(a) > 1 and a[1] == 'true' else 'dense'
    if 'wide::fwd' in n:                                    # <XC, BIG, HEAD>
        a += ['false'] * (3 - len(a))
        return 'compact_x' if a[0] == 'true' else 'head_only_out' if a[2] == 'true' else 'dense'
    if 'grand_fwd' in n:                                    # <C, XC>; head-only output: run-time argument
        return 'compact_x' if len(a) > 1 and a[1] == 'true' else None
    return None


def main():
    root, workload = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else 'profiles/pmc.json'
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row.get('Kernel_Name', '')
            k = name(kn)
            if not k:
                continue
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
            v = variant(kn)
            if v:
                acc[k + ':' + v][row['Counter_Name']].append(float(row['Counter_Value']))
    res = {}
    for k, c in sorted(acc.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        r = {}
        if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
            r['traffic_bytes'] = int((2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024)
            r['fetch_kib'], r['write_kib'] = round(m['FETCH_SIZE'], 1), round(m['WRITE_SIZE'], 1)
        if 'GRBM_GUI_ACTIVE' in m:
            cyc = m['GRBM_GUI_ACTIVE'] / N_XCD
            r['kernel_cycles'] = round(cyc)
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
                r['mfma_util'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (N_SIMD * cyc), 4)
                r['mfma_insts'] = round(m.get('SQ_INSTS_MFMA', 0))
            if 'SQ_ACTIVE_INST_VALU' in m:
                r['valu_util'] = round(4 * m['SQ_ACTIVE_INST_VALU'] / (N_SIMD * cyc), 4)
        if 'SQ_WAIT_ANY' in m and 'SQ_WAVE_CYCLES' in m:
            r['wait_frac'] = round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 4)
            r['issue_stall_frac'] = round(m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES'], 4)
        if 'TCC_HIT_sum' in m and 'TCC_MISS_sum' in m:
            r['l2_hit'] = round(m['TCC_HIT_sum'] / max(m['TCC_HIT_sum'] + m['TCC_MISS_sum'], 1), 4)
        if 'SQ_LDS_BANK_CONFLICT' in m and 'SQ_LDS_IDX_ACTIVE' in m:
            r['lds_conflict_frac'] = round(m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1), 4)
        res[k] = r
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[workload] = res
    json.dump(data, open(out, 'w'), indent=1, sort_keys=True)
    traffic = {wl: {k: r['traffic_bytes'] for k, r in ks.items() if 'traffic_bytes' in r} for wl, ks in data.items()}
    json.dump(traffic, open(os.path.join(os.path.dirname(out) or '.', 'traffic.json'), 'w'), indent=1, sort_keys=True)
    for k, r in res.items():
        print(workload, k, r)


if __name__ == '__main__':
    main()
