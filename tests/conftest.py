import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))     # oracle/ is test infrastructure, importable only from here
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def real_proofs():
    return load_golden('real_proofs.json')


@pytest.fixture(scope='session')
def verify_corpus():
    return load_golden('verify_corpus.json')


@pytest.fixture(scope='session')
def precompile_kats():
    return load_golden('precompile_kats.json')


@pytest.fixture(scope='session')
def revert_vectors():
    return load_golden('revert_bytes.json')


@pytest.fixture(scope='session')
def wire_cases():
    return load_golden('wire_cases.json')


@pytest.fixture(scope='session')
def g2_membership_points():
    return load_golden('g2_membership_points.json')['points']
