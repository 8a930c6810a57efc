"""The C++ host mirror of the reference traits compiles against include/zkv.h, links the C ABI and reproduces the host logic
(CPU part) and, on a GPU box, the verify outcomes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(real_proofs):
    from stylus_zkvm_verifiers_amd import build
    build.build(verbose=False)
    src = os.path.join(ROOT, 'tests', 'host_cpp', 'test_host_mirror.cpp')
    exe = os.path.join(ROOT, 'tests', 'host_cpp', 'test_host_mirror')
    libdir = os.path.join(ROOT, 'stylus_zkvm_verifiers_amd')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-o', exe, src, '-L' + libdir, '-lzkv_mi355x', '-Wl,-rpath,' + libdir])
    r = real_proofs['risc0']
    out = subprocess.check_output([exe, r['control_root'], r['bn254_control_id'], r['seal'], r['image_id'], r['journal_digest']],
                                  timeout=300).decode()
    return dict(kv.split('=', 1) for kv in out.split())


def test_cpp_mirror_host_logic(real_proofs):
    kv = _build_and_run(real_proofs)
    r = real_proofs['risc0']
    assert kv['initialized0'] == '0' and kv['init'] == '1'
    assert kv['reinit_status'] == '3' and kv['reinit_err'] == '0dc149f0'
    assert kv['selector'] == r['selector'] and kv['vk_digest'] == r['vk_digest']
    assert kv['sp1_version'] == 'v5.0.0'
    assert kv['calldata_len'] == '8452' and kv['calldata_head'] == 'f8b3b60b' + '00' * 31 + '60'
    assert (kv['multi_shards'], kv['multi_selector'], kv['multi_initialized'], kv['sp1_multi_shards'], kv['empty_mask']) == ('1', r['selector'], '1', '1', 'refused')
    if 'verify_runtime_error' in kv:
        assert kv['verify_runtime_error'] == '-2'       # ZKV_ERR_NO_DEVICE: no CPU fallback


@pytest.mark.gpu
def test_cpp_mirror_verifies_on_gpu(real_proofs):
    kv = _build_and_run(real_proofs)
    assert kv['verify_ok'] == '1' and kv['multi_verify_ok'] == '1'
    assert kv['mismatch_status'] == '5'
    assert kv['mismatch_err'].startswith('b8b38d4c9e39696c')
    # wire layer through the C++ mirror: verify() call -> true word, getSelector() -> bytes4 word, garbage -> empty revert
    assert kv['call0_reverted'] == '0' and kv['call0_ret'] == '00' * 31 + '01'
    assert kv['call1_ret'] == real_proofs['risc0']['selector'] + '00' * 28
    assert (kv['call2_reverted'], kv['call2_len'], kv['call2_status']) == ('1', '0', '6')
