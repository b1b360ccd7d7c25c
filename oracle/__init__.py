"""CPU oracle for the BOA hot path -- TEST INFRASTRUCTURE ONLY.

A numpy / scipy / torch-CPU restatement of the reference's algorithm for the path named in
BASELINE.json (`north_star`): nnU-Net sliding-window inference + TotalSegmentator label merge +
BOA body-composition / HU aggregations.  Every function cites the reference file:line it follows
(paths relative to the reference root; NN/ = body_organ_analysis/_external/nnunetv2, TS/ =
.../totalsegmentator, BCA/ = .../body_composition_analysis, BOA/ = body_organ_analysis).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package, and only as the checker / the reported CPU baseline.  The product
(`body-and-organ-analysis_amd/`) must never import it and fails loudly without its HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * pinned against the reference itself, executed in the dev container through the stub-import
    harness (`tests/golden/make_golden.py` -> `tests/golden/*.npz|json`): tile starts, Gaussian map,
    sliding-window fp16 accumulation, CTNormalization, argmax, label merge, resampling,
    tissue subclassification, slice-wise / aggregated BCA measurements, per-label HU metrics,
    model/device resolution.
  * PARITY UNPINNED (third-party code absent from /root/reference and from this image):
      - the network itself (`dynamic_network_architectures==0.4.3` PlainConvUNet): restated from
        torch.nn following the kwargs the reference synthesises (NN/utilities/plans_handling/
        plans_handler.py:59-92); checked only against torch-CPU ops.
      - skimage / cv2 morphology (binary_erosion with padded footprint, measure.label,
        remove_small_objects, findContours fill): restated with scipy.ndimage.
"""
