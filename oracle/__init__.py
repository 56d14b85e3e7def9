"""CPU oracle for the AtlasPatch hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product (``atlaspatch_amd``) never
imports this package and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * geometry / contour scaling / grid scan / H5 row order / extract_batch
    control flow: PINNED -- golden vectors under ``tests/golden`` were produced
    by running the reference's own modules (imported from /root/reference with
    stub packages, see ``tests/golden/gen_golden.py``).
  * ViT forward: PINNED against the reference's ``PatchFeatureExtractor`` driving
    a seeded HF ``ViTModel`` (same script).
  * The five OpenCV primitives the path calls (findContours, contourArea,
    pointPolygonTest, boundingRect + constants) live in opencv-python
    (``>=4.7.0``, unpinned in the reference's pyproject.toml:35) which is absent
    from this image: ``cv2_restated.py`` restates the published algorithms
    (Suzuki-Abe border following as implemented by OpenCV) -> PARITY UNPINNED
    for those primitives themselves.
"""
