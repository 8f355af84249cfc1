"""CPU oracle of the Ctrl-Adapter denoising hot path (ControlNet + Ctrl-Adapter + router).

THIS PACKAGE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import it -- as the checker, never as the thing measured or shipped.  The product path
(ctrl-adapter_amd/) never imports it and fails loudly when the HIP library is missing.

What it is: a plain-PyTorch fp32 restatement of the reference algorithm, module tree and state-dict keys
identical to the reference so real checkpoints would load:
  oracle/blocks.py      diffusers v0.27.x building blocks (the un-vendored third-party dependency that holds
                        the arithmetic; pinned by the reference's own header comments to v0.27.2 /
                        commit 2dcf64b7) restated from the published source
  oracle/controlnet.py  controlnet/controlnet.py:62-104,179-438,662-881 and controlnet/multicontrolnet.py:45-99
  oracle/adapter.py     model/ctrl_adapter.py:17-224, model/adapter_spatial_temporal.py:11-292,
                        model/resnet_block_2d.py:61-221
  oracle/router.py      model/ctrl_router.py:9-112 and the caller-side merge
                        (i2vgen_xl/pipelines/i2vgen_xl_controlnet_adapter_pipeline.py:1000-1022, train.py:1262-1276)
  oracle/image_prep.py  model/ctrl_helper.py:268-296 (prepare_images) incl. Pillow's 8-bit Lanczos resampling in numpy --
                        PINNED bit for bit against Pillow itself (tests/test_image_prep.py)

PARITY PINNING STATUS: *partially pinned*.  The reference has no tests, golden vectors or fixtures for this
path, and `diffusers` is not installed here, so the reference cannot be imported as is.
What IS pinned: the reference's OWN files (model/*.py, controlnet/*.py) are executed unmodified from
/root/reference on top of oracle/_shim (a minimal `diffusers` package whose blocks are oracle/blocks.py), by
tests/golden/make_golden.py; their outputs are committed under tests/golden/ (digests of every output, whole fp32 tensors
for a selection of every case) and the standalone restatements in this package are checked against them.  What is NOT pinned: the diffusers v0.27.x block
arithmetic in oracle/blocks.py (restated from the published source; cross-checked only against independent
naive formulations in tests/test_oracle_blocks.py).
"""
