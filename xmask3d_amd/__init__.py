"""xmask3d_amd - MI355X (gfx950) native hot path of XMask3D.

Layout:
  csrc/            HIP kernels + the C ABI (include/xm3d.h) -> libxm3d_hip.so
  _lib.py, ops.py  ctypes binding and tensor-level wrappers (no CPU fallback)
  me_compat.py     MinkowskiEngine-compatible operator surface (drop-in seam)
  msda.py          MultiScaleDeformableAttention-compatible surface
  voxelizer.py     Voxelizer mirror (dataset/voxelizer.py) on the GPU
  mink_unet.py, pc_processor.py   sparse 3D backbone definitions
  synthetic.py     ScanNet-shaped synthetic scenes + point->pixel mapping
"""
__version__ = "0.1.0"
