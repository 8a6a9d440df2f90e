"""casualhdrsplat_amd -- MI355X-native differentiable 3D Gaussian rasterizer for HDR splatting.

Scope: the rasterizer hot path only, behind the
GaussianRasterizer / GaussianRasterizationSettings API.  Compute lives in
casualhdrsplat_amd/libhdrsplat.so (hand-written HIP, gfx950) reached through the C ABI of
include/hdrsplat.h; importing the package does not load the library, calling it does, and a
missing library is a hard error (no CPU fallback).
"""
from .rasterizer import (BinningOverflow, DensifyStats, GaussianRasterizationSettings, GaussianRasterizer,
                         SortChainStalled, inspect_state, rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "DensifyStats", "BinningOverflow", "SortChainStalled",
           "rasterize_gaussians", "inspect_state"]
__version__ = "0.1.0"
