"""CPU oracle for the SPNet detection hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from this package; the product (``spnet_amd/``) never does and fails loudly when its HIP
library is missing instead of falling back to this code.

Contents
--------
``numpy_ref``   numpy restatement of the reference's host arithmetic: grid codec, ``my_loss`` /
                ``custom_loss`` (+ closed-form gradient), 1-cycle LR table, Keras-form Adam,
                cutout / salt-and-pepper with the reference's RNG call order, decode, count metrics.
                PINNED: checked against ``tests/golden/reference_numpy.npz``, which was produced
                by the reference's own functions (``tests/golden/make_goldens.py``).
``torch_ref``   torch-CPU restatement of the network the reference builds through Keras 2.1.3 /
                TensorFlow 1.14 (stem + Xception + Dense head, TF-SAME padding, Keras BatchNorm and
                Adam semantics).  PARITY UNPINNED at the Keras boundary: Keras/TensorFlow are not
                installable in the build container and the reference ships no golden vectors for
                the network, so this part is anchored on structural known-answers only (parameter
                counts 50,353,481 / 50,298,935 / 54,546 and shapes (165,165,3) / (5,5,2048) from
                paper/run_logs/log_DatasetA...txt:94-101).
``warp_ref``    numpy restatement of the OpenCV warps used by augment_preproc.py (flip / rotate /
                translate).  PARITY UNPINNED (OpenCV is absent; bilinear weights are exact float,
                not cv2's 5-bit fixed point).
"""
