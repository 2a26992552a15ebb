"""oracle/ — CPU restatement of the reference's Smith-Waterman path + handle on the real
reference build.  TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, never from the product package."""
