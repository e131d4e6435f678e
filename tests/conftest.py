import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """The shared library is a build artefact (git-ignored): make sure the tests see one that matches
    the sources.  Building is not a fallback - without hipcc and without a library the GPU tests fail."""
    try:
        from lumfuncmcmc_amd import build
        if build.is_stale():
            build.build_library(verbose=False)
    except Exception as e:                      # no hipcc here: leave it to the tests to say so
        sys.stderr.write("conftest: could not (re)build liblfmcmc.so: %s\n" % e)
