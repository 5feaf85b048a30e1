"""bench.py's readers of the committed PMC passes (no GPU): round-5 verdict, weak 3 -- `roofline.traffic` skipped conv_gemm_split_kernel because
of a regular expression; both readers must count every GEMM kernel family of the file."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _write(d, name, rows):
    with open(os.path.join(d, name), 'w') as f:
        for r in rows:
            f.write(r + '\n')


def test_pmc_readers_count_every_gemm_family(tmp_path):
    import bench
    root = str(tmp_path)
    d = os.path.join(root, 'profiles', 'r9')
    os.makedirs(d)
    fam = [('conv_gemm_kernel', 100), ('conv_gemm_split_kernel', 60), ('conv_gemm_ks_kernel', 40), ('ew_chain_kernel_v4', 10)]
    _write(d, 'pmc_FETCH_SIZE.txt', ['%-28s dispatches %6d  FETCH_SIZE=%g' % (k, n, n * 1000.0) for k, n in fam])
    _write(d, 'pmc_WRITE_SIZE.txt', ['%-28s dispatches %6d  WRITE_SIZE=%g' % (k, n, n * 500.0) for k, n in fam])
    _write(d, 'pmc_mfma.txt', ['%-28s dispatches %6d  GRBM_GUI_ACTIVE=%g  SQ_BUSY_CYCLES=1  SQ_INSTS_MFMA=1  SQ_VALU_MFMA_BUSY_CYCLES=%g  SQ_WAVE_CYCLES=1' % (
        k, n, 8.0 * n * 1000, 0.5 * n * 1000 * 1024) for k, n in fam[:3]])
    t = bench.pmc_traffic(root, '')
    # 200 GEMM dispatches: (2 x 1000 + 500) KB each; a reader that skipped the split kernel would still read 2500 KB -- so make the families differ
    assert abs(t['traffic'] - 2500 * 1024) < 1e-6 * 2500 * 1024
    _write(d, 'pmc_FETCH_SIZE.txt', ['%-28s dispatches %6d  FETCH_SIZE=%g' % (k, n, n * (3000.0 if 'split' in k else 1000.0)) for k, n in fam])
    t = bench.pmc_traffic(root, '')
    want = (2.0 * (100 * 1000 + 60 * 3000 + 40 * 1000) / 200 + 500.0) * 1024
    assert abs(t['traffic'] - want) < 1e-6 * want, (t, want)
    m = bench.pmc_mfma(root, '', 20.0, 10.0, 2.0)
    assert abs(m['mfma_util_serial_pmc'] - 0.5) < 1e-9 and set(m['mfma_util_serial_pmc_by_kernel']) == {'conv_gemm_kernel', 'conv_gemm_split_kernel', 'conv_gemm_ks_kernel'}
    # 200 dispatches / 20 per step = 10 steps; busy per step = 0.5 * 200 * 1000 * 1024 / 10; step cycles = 10 ms x 2 GHz x 1024
    assert abs(m['mfma_util_timed_estimate'] - (0.5 * 200 * 1000 * 1024 / 10) / (10e-3 * 2e9 * 1024)) < 1e-12


def test_committed_profiles_parse():
    """The readers on the repository's own committed profiles: a figure, from the newest round that has the files."""
    import bench
    t = bench.pmc_traffic(ROOT, '')
    m = bench.pmc_mfma(ROOT, '', 303.0, 21.0, 2.4)
    assert t.get('traffic', 0) > 0 and re.match(r'profiles/r\d+/', t['traffic_source'])
    assert 0.0 < m['mfma_util_serial_pmc'] < 1.0
