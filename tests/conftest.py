import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _host_cpus() -> int:
    """CPUs the container grants (affinity cut by the cgroup quota; bench.py: host_cpus has the story): thread pools sized to the
    256 CPUs a GPU box SHOWS spend its 16-CPU quota spinning and the kernel parks the whole suite for the rest of each 100 ms."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f: quota, period = f.read().split()[:2]
        if quota != 'max': n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError): pass
    return max(1, n)


for _var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):      # before torch / numpy build their pools
    os.environ.setdefault(_var, str(_host_cpus()))
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run through gpurun / at round end)')
    # a fresh checkout has no built artefacts (they are git-ignored): build the HIP library and the
    # oracle once, exactly as the driver's build() check does
    needed = [os.path.join(ROOT, 'ppq_amd', 'libppq_hip.so'), os.path.join(ROOT, 'oracle', 'libppq_oracle.so')]
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()


# Collection order (the driver runs `pytest -x`): kernel-level bit-exact parity first, then the unmodified reference
# on these kernels, then the host layer, and the training-based passes LAST, so that no single test further down the
# stack can hide the parity suite (round 2: one optimizer-outcome threshold masked 234 tests).
_ORDER = ['test_oracle_golden.py', 'test_oracle_ref_kernels.py', 'test_host_cpu.py', 'test_gpu_kernels.py', 'test_gpu_fp8_reference.py',
          'test_gpu_kernels_reference.py', 'test_gpu_reference.py',
          'test_gpu_plugin_seam.py', 'test_gpu_calibration.py', 'test_gpu_rccl.py', 'test_gpu_finetune.py']


def pytest_collection_modifyitems(config, items):
    """Fixed file order (above); GPU tests are skipped (not failed) when no device is visible."""
    rank = {name: i for i, name in enumerate(_ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), len(_ORDER) - 1.5))      # stable: in-file order kept
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


# ---- the reference-dependent parity tests ride on git-ignored build artefacts (oracle/_ref/: the reference's kernels compiled for the
# host + a staged importable copy of its Python, made by build() only where /root/reference exists).  Where they are absent those tests
# SKIP -- and a green suite must not hide that: the header says what is staged, the summary counts the skips, and
# PPQ_REQUIRE_REFERENCE=1 turns every such skip into a failure.
_REF_WORDS = ('reference', '_ref')
_ref_skips = []


def _reference_state():
    ref = os.path.join(ROOT, 'oracle', '_ref')
    return {'staged python (oracle/_ref/ppq_stage)': os.path.isdir(os.path.join(ref, 'ppq_stage', 'ppq')),
            'libref_kernels.so': os.path.exists(os.path.join(ref, 'libref_kernels.so')),
            'libref_common.so': os.path.exists(os.path.join(ref, 'libref_common.so')),
            'libref_hist_mse.so': os.path.exists(os.path.join(ref, 'libref_hist_mse.so'))}


def pytest_report_header(config):
    st = _reference_state()
    req = os.environ.get('PPQ_REQUIRE_REFERENCE', '0') not in ('', '0')
    return ['reference staged: ' + ('yes' if all(st.values()) else 'NO') + ' (' + ', '.join(f'{k}: {"yes" if v else "no"}' for k, v in st.items()) + ')'
            + f'; PPQ_REQUIRE_REFERENCE={"1 (a skip for want of the reference FAILS)" if req else "0 (such tests skip; counted in the summary)"}']


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    if not rep.skipped or getattr(rep, 'wasxfail', None) is not None: return
    reason = rep.longrepr[2] if isinstance(rep.longrepr, tuple) and len(rep.longrepr) == 3 else str(rep.longrepr)
    if 'no GPU visible' in reason or not any(w in reason for w in _REF_WORDS): return
    _ref_skips.append(item.nodeid)
    if os.environ.get('PPQ_REQUIRE_REFERENCE', '0') not in ('', '0'):
        rep.outcome = 'failed'
        rep.longrepr = f'PPQ_REQUIRE_REFERENCE=1 and this parity test could not run: {reason}'


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if _ref_skips:
        terminalreporter.write_line(f'reference NOT staged: {len(_ref_skips)} reference-parity tests did not run (build where /root/reference '
                                    f'exists, or set PPQ_REQUIRE_REFERENCE=1 to make this an error)', yellow=True)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


def record_parity_residue(kind: str, test: str, **detail) -> None:
    """every admitted relaxation of a parity statement (a KL near-tie, a vendor-convolution non-repeatability
    that made a comparison inconclusive) is appended, per run, to gpurun_out/parity_residue.jsonl -- the builder copies the
    round's file to profiles/.  Never raises."""
    import json
    import time
    try:
        out = os.path.join(os.environ.get('GRAFT_REPO_ROOT', ROOT), 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_residue.jsonl'), 'a') as f:
            f.write(json.dumps({'time': time.strftime('%Y-%m-%d %H:%M:%S'), 'kind': kind, 'test': test, **detail}) + '\n')
    except Exception:
        pass
