"""Regenerates tests/golden/cornell_box.npz from the reference's Cornell box asset (a public-domain
data file, assets/cornell_box/CornellBox-Original.{obj,mtl}) with nova_amd.scene.load_obj.
Run in the build container only: /root/reference does not exist on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from nova_amd import scene as S

v, m, e, _ = S.load_obj("/root/reference/assets/cornell_box/CornellBox-Original.obj")
np.savez_compressed(os.path.join(os.path.dirname(__file__), "cornell_box.npz"),
                    verts=v, meta=m.view(np.uint8).reshape(m.shape[0], -1), emissive=e)
print("cornell_box.npz:", v.shape[0], "triangles")
