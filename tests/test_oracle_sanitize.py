"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on this pool, so the
sanitized build is CPU only): a slice of the golden corpus must run clean and give the golden statuses."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def driver():
    exe = os.path.join(ROOT, 'oracle', 'sanitize_driver_asan')
    srcs = [os.path.join(ROOT, 'oracle', 'zkv_oracle.c'), os.path.join(ROOT, 'oracle', 'sanitize_driver.c')]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(['gcc', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-fno-omit-frame-pointer',
                               '-fvisibility=default', '-o', exe] + srcs)
    return exe


def test_corpus_slice_runs_clean(driver, verify_corpus):
    ctx = verify_corpus['risc0_ctx']
    picks = [c for c in verify_corpus['cases'] if c['name'] in (
        'real proof', 'rerandomised 0', 'flip bit in C.x', 'A = (0,Q) -> wraps to infinity under negate_g1', 'B out of subgroup (on twist)',
        'B = infinity', 'wrong selector', 'len 3', 'len 261', 'public values 55 bytes', 'vkey >= R', 'A = (0,0)')]
    assert len(picks) >= 12
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='halt_on_error=1')
    for c in picks:
        if c['vm'] == 'risc0':
            args = ['risc0', ctx['control_root'], ctx['bn254_control_id'], c['seal'] or '', c['image_id'], c['journal_digest']]
        else:
            args = ['sp1', c['vkey'], c['public_values'], c['proof']]
        p = subprocess.run([driver] + args, capture_output=True, env=env, timeout=120)
        assert p.returncode == 0, (c['name'], p.stderr.decode()[-2000:])
        assert b'runtime error' not in p.stderr and b'AddressSanitizer' not in p.stderr, c['name']
        assert int(p.stdout.strip()) == c['status'], c['name']


def test_wire_layer_decoding_runs_clean(driver, wire_cases):
    """The calldata decoder of the oracle under ASan / UBSan on damaged calldata (truncations, huge lengths, bad offsets):
    no over-reads, and the golden return / revert data."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from wire_util import calldata_of
    import oracle_lib as ol
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='halt_on_error=1')
    ctx = wire_cases['risc0_ctx']
    picks = [c for c in wire_cases['cases'] if c['reverted'] and c['status'] in (6, 4) or c['method'] == 'raw'][:60]
    picks += [c for c in wire_cases['cases'] if c['name'].startswith('real proof')]
    assert len(picks) >= 40
    for c in picks:
        cd = calldata_of(c, ol.risc0_encode_call, ol.risc0_encode_call, ol.sp1_encode_call).hex()
        if c['vm'] == 'risc0':
            args = ['call_risc0'] + ([ctx['control_root'], ctx['bn254_control_id']] if c['ctx'] == 'init' else ['-', '-']) + [cd]
        else:
            args = ['call_sp1', cd]
        p = subprocess.run([driver] + args, capture_output=True, env=env, timeout=120)
        assert p.returncode == 0, (c['name'], p.stderr.decode()[-2000:])
        assert b'runtime error' not in p.stderr and b'AddressSanitizer' not in p.stderr, c['name']
        rev, _, ret = p.stdout.decode().strip().partition(' ')
        assert (rev == '1', ret) == (c['reverted'], c['returndata']), c['name']
