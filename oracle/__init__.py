"""TEST INFRASTRUCTURE ONLY.  CPU restatements of the reference algorithms (the checker the HIP path
is compared with).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; vilco_amd/ never does."""
