"""Evidence hygiene: every `roofline*` field of the committed bench line can be recomputed from tracked files
under profiles/ alone, and the two counter summaries come from one pass set.  No GPU."""
import csv
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_traffic_json_is_the_traffic_column_of_pmc_json():
    pmc = json.load(open(os.path.join(PROF, 'pmc.json')))
    traffic = json.load(open(os.path.join(PROF, 'traffic.json')))
    assert set(pmc) == set(traffic)
    for wl, kernels in pmc.items():
        want = {k: r['traffic_bytes'] for k, r in kernels.items() if 'traffic_bytes' in r}
        assert traffic[wl] == want, wl


def _latest_bench_line():
    """(path, parsed line) of the newest committed default-workload bench line that carries the round-3 fields."""
    for path in sorted(glob.glob(os.path.join(PROF, 'r*_bench.json')), reverse=True):
        txt = open(path).read().strip()
        if not txt:
            continue
        line = json.loads(txt.splitlines()[-1])
        if isinstance(line.get('roofline'), dict) and 'profile' in line['roofline']:
            return path, line
    return None, None


def test_bench_line_roofline_is_reproducible_from_profiles():
    path, line = _latest_bench_line()
    if line is None:
        pytest.skip("no committed bench line with the round-3 roofline fields yet")
    from make_pmc_json import name as kname, variant as kvariant
    rf = line['roofline']
    wl = line['config']['workload']
    # frac follows from the line's own raw duration and byte count ...
    assert rf['frac'] == pytest.approx(rf['alg_bytes_per_launch'] / (rf['avg_launch_us'] * 1e-6) / 8e12, rel=2e-3)
    assert rf['avg_launch_us'] >= rf['avg_launch_us_net_of_overhead']          # nothing subtracted from what prices frac
    # ... the byte count from SURVEY.md 8(d): 8 N C + 4 (E + N + 1) for a dense target-pass launch of the metric workload ...
    import bench
    w = bench.WORKLOADS[wl]
    assert rf['kernel'] == 'backward_target' and rf['variant'] == 'dense'
    if wl == 'poisson2d_64x64_b32_L4_C64':
        n_nodes, n_edges = 32 * 4096, 32 * 23564                              # n = 64 -> N = 4096, E = 23 564 per mesh
        assert rf['alg_bytes_per_launch'] == bench.algorithmic_bytes('backward_target', n_nodes, n_edges, w['hidden'], 0) == 70649348
    # ... the committed rocprofv3 summary gives the same duration (the event pair adds its dispatch share: <= 12 % above, never below)
    prof = rf['profile']
    assert prof is not None, "no committed rocprofv3 kernel stats for the bench workload"
    tot = calls = 0
    for row in csv.DictReader(open(os.path.join(ROOT, prof['file']))):
        if kname(row['Name']) == rf['kernel'] and kvariant(row['Name']) == rf['variant']:
            tot += float(row['TotalDurationNs']); calls += int(row['Calls'])
    assert calls > 0
    csv_us = tot / calls / 1e3
    assert prof['avg_us'] == pytest.approx(csv_us, rel=1e-3)
    assert prof['frac'] == pytest.approx(rf['alg_bytes_per_launch'] / (csv_us * 1e-6) / 8e12, rel=2e-3)
    assert rf['frac'] <= prof['frac'] * 1.03, "the line's frac may not exceed what the rocprofv3 summary supports"
    assert rf['frac'] >= prof['frac'] * 0.85
    # traffic is the committed counter summary's entry for that kernel variant
    pmc = json.load(open(os.path.join(PROF, 'pmc.json')))[wl]
    assert rf['traffic'] == pmc[f"{rf['kernel']}:{rf['variant']}"]['traffic_bytes']
    # the matrix-pipe figure is the measured one
    rm = line['roofline_mfma']
    assert rm['frac'] == rm['measured']['hot_kernels_time_weighted'] and 'reference_flops_vs_fp32_peak' in rm


def _newest_parity_record():
    paths = sorted(glob.glob(os.path.join(PROF, 'r*_parity.json')))
    return paths[-1] if paths else None


def test_parity_margins_are_committed_for_every_case():
    """the parity run's margins are on record - `profiles/rNN_parity.json` (written by tests/test_gpu_parity.py on
    the GPU box, copied here) has an entry for every case of the parity matrix in every kernel dispatch it runs in, each parameter
    with its error against fp64, the oracle's single-run noise, the band where one was computed, and the rule that admitted it."""
    path = _newest_parity_record()
    if path is None:
        pytest.skip("no profiles/rNN_parity.json committed yet (GPU run pending)")
    log = json.load(open(path))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_gpu_parity as tp
    missing = [i for i in tp.PARAM_IDS if i not in log['cases']]
    assert not missing, missing
    for cid, rec in log['cases'].items():
        assert rec['coordinates']['normwise'] <= log['coord_tol'], cid
        for row in rec['gradients']:
            assert row['passed'] and row['rule'] in ('floor', 'noise', 'band'), (cid, row)
            assert row['e64'] <= row['bound'], (cid, row)
            if row['rule'] == 'band':                        # only in a case whose oracle is itself noisy on some parameter
                assert row['noisy_case'] and max(r['noise_single_run'] for r in rec['gradients']) >= 0.5 * log['grad_tol'], (cid, row)


def test_parity_record_is_not_older_than_the_kernels():
    """the committed parity record must come from a run of the kernels as they are: it carries a fingerprint of
    `g_adaptivity_amd/csrc/` + `include/gadapt_hip.h` taken on the GPU box when it was written; any later change of a kernel source
    fails this test until tests/test_gpu_parity.py has run again and `gpurun_out/<round>_parity.json` is committed to `profiles/`."""
    path = _newest_parity_record()
    if path is None:
        pytest.skip("no parity record committed yet")
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_gpu_parity as tp
    log = json.load(open(path))
    assert log.get('kernel_sources_sha16') == tp.kernel_sources_sha16(ROOT), (
        f"{os.path.basename(path)} was recorded with other kernel sources than this tree's: re-run tests/test_gpu_parity.py on the GPU and "
        "commit gpurun_out/<round>_parity.json to profiles/")
