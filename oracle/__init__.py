"""ORACLE -- test infrastructure only.

A CPU restatement of the reference's (MMOCKING/RIDERS) algorithm for the RC-Net / Scale-Map-Learner hot path,
used ONLY as the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
riders_amd/ imports this package; the product path fails loudly when the HIP library is missing.
"""
