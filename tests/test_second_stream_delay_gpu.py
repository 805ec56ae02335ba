"""The second-stream equality tests once more in a child process whose second stream is held back by ~2 ms before every weight
gradient (``SPK_WGRAD_STREAM_DELAY``, read by ``ops.side_stream_delay``): a consumer that does not wait for the stream -- an add by
the autograd engine, a gradient hook, the optimizer -- would now certainly read an unfinished gradient."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_equality_tests_with_the_second_stream_held_back():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPK_WGRAD_STREAM_DELAY="5000000")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-q", "-x", "-k", "second_stream_equal", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "4 passed" in r.stdout, r.stdout[-1500:]
