"""TEST INFRASTRUCTURE — CPU oracle of the Metarank /rank hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under metarank_amd/ imports it.  See the headers of forest_oracle.cpp and
assembly_oracle.cpp for the reference file:line each function restates and for the parity
pinning status.
"""
