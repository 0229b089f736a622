"""TEST INFRASTRUCTURE ONLY -- never imported by the product package `gan_amd`.

CPU restatement (plain PyTorch, functional, state-dict driven) of the reference's
StyleGAN2 training hot path, used as the parity checker by `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`.

Parity status: pinned.  `oracle/make_goldens.py` imports the reference itself
(/root/reference, CPU, with the harness-side patches of SURVEY.md section 8c) and
writes input/output vectors to `tests/golden/`; `tests/test_oracle_golden.py`
checks every function here against those vectors.
"""
