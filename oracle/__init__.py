"""oracle — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's Monte-Carlo variational forward path (yliess86/BayeFormers).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product (bayeformers_amd)
never does, and fails loudly without its HIP extension instead of falling back to anything here.
"""
