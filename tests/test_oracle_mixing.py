"""CPU: the mixing oracle against the reference's own generate_noisy_wav (build container only: needs /root/reference; the reference module
imports librosa / soundfile at import time, which are absent, so only its function body is executed, from its source text)."""
import ast
import os

import numpy as np
import pytest

from oracle.mixing import generate_noisy_wav

REF = "/root/reference/generate_noisy_data.py"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference not present (GPU box)")
def test_mixing_oracle_equals_reference_function():
    tree = ast.parse(open(REF).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "generate_noisy_wav")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    rng = np.random.default_rng(0)
    speech = (rng.standard_normal(4000) * 0.1 + 0.01)
    noise = rng.standard_normal(20000) * 0.3 - 0.02
    for snr, start in ((0, 17), (5, 9000), (-5, 15999)):
        orig = np.random.randint
        np.random.randint = lambda lo, hi: start
        try:
            ref = ns["generate_noisy_wav"](speech, noise, snr)
        finally:
            np.random.randint = orig
        assert np.array_equal(ref, generate_noisy_wav(speech, noise, snr, start))
