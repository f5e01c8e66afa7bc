"""CPU oracle for the GVL deformable-attention hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (gvl_amd/) never imports it; see DESIGN.md section 3.
"""
