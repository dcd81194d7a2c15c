"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from this package; the product package (image_restoration_platform_amd)
never does.  See each module's header for what pins it.
"""
