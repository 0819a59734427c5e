"""CPU oracle for the DCCRN / CRN / FullSubNet training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch(CPU)/NumPy restatement of
the reference algorithm (seorim0/DNN-based-Speech-Enhancement-in-the-frequency-domain),
written from SURVEY.md section 8 and the reference sources cited in each docstring.
It is the *checker*: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it.  Nothing in the product package
(`dnn-based-speech-enhancement-in-the-frequency-domain_amd/`) imports it and the product
has no CPU fallback - it raises if the HIP library is missing.

Parity pinning: every function here is checked in `tests/test_oracle_golden.py` against
golden vectors captured by importing the real reference in the build container
(`tests/golden/make_golden.py`, outputs committed as `tests/golden/*.npz`) and against the
known-answer values recorded in SURVEY.md Appendix A (Q4, Q7, Q15).
PMSQE (third-party `asteroid`, absent here) is NOT restated: parity unpinned.
"""
