"""Synthetic workloads for tests, tools and bench.py: Ranklens-shaped feature state / requests and
forests / encoders written in the real on-disk formats.  Test and measurement infrastructure - not part of the product
package (metarank_amd/), which never imports it."""
