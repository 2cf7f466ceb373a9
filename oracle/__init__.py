"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's VFM-segmentation hot path (oracle/torch_ref.py),
the shim that imports the real reference files in the build container
(oracle/ref_shim.py) and the golden-vector generator (oracle/gen_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (vfmseg_amd) never does.
"""
