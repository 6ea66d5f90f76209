"""Builds tests/golden/indoor1.npz from a DATA file the reference ships (run in the build container only):

  examples/indoor1.png      427 x 423 (W x H) RGBA: the photograph BASELINE.json configs[1] names ("examples/indoor1.png 512x512, --model_name=pos_mlp
                            --opt_order='rm a' --opt_env_from=2")

Only its pixels are stored (uint8 RGBA, as decoded); no reference code.  The reference ships no MaterialNet prediction and no result for this
photograph (output_imgs/ holds indoor2's and jinjya's), so the test that uses it (tests/test_gpu_configs.py::test_config1_as_written_*) starts
from the flat prior and checks the run, not where it lands.
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    im = np.asarray(Image.open("/root/reference/examples/indoor1.png"), dtype=np.uint8)
    assert im.shape == (423, 427, 4), im.shape      # 423 rows x 427 columns
    np.savez_compressed(os.path.join(HERE, "indoor1.npz"), image_rgba_u8=im)
    print(im.shape, im.dtype, "alpha range", im[..., 3].min(), im[..., 3].max())


if __name__ == "__main__":
    main()
