"""Writes tests/golden/tower_witness.json.

The vectors below are the literal inputs and expected outputs of the reference's own
known-answer tests over GoldilocksExt2 (ceno_zkvm/src/scheme/utils.rs:934-1194):
  test_infer_tower_witness, test_interleaving_mles_to_mles,
  test_interleaving_mles_to_mles_padding (2 cases), test_interleaving_mles_to_mles_edgecases,
  test_infer_tower_logup_witness.
They are data (values asserted by the reference tests), transcribed by hand; every value is an
embedded base-field element `E::from_u64(v)` = [v, 0].  The reference cannot be run in this
image (Rust, no toolchain), so the file is not produced by executing it.
"""
import json
import os

golden = {
    "field": {"p": 0xFFFFFFFF00000001, "ext_w": 7, "note": "all vectors use base-embedded values [v,0]"},
    "infer_tower_product_witness": [
        {  # utils.rs:934-965
            "ref": "ceno_zkvm/src/scheme/utils.rs:934-965",
            "num_vars": 2,
            "last_layer": [[1, 2], [3, 4]],
            "num_layers": 2,
            "limbs_per_layer": 2,
            # left[0]*right[0] of layer 0 == product of all leaves
            "final_product": 1 * 2 * 3 * 4,
        }
    ],
    "interleaving_mles_to_mles": [
        {"ref": "utils.rs:967-998", "mles": [[1, 2], [3, 4], [5, 6], [7, 8]], "num_instances": 2, "num_limbs": 2,
         "default": 1, "expected": [[1, 3, 5, 7], [2, 4, 6, 8]]},
        {"ref": "utils.rs:1005-1031 (limb level padding)", "mles": [[1, 2], [3, 4], [5, 6]], "num_instances": 2,
         "num_limbs": 2, "default": 0, "expected": [[1, 3, 5, 0], [2, 4, 6, 0]]},
        {"ref": "utils.rs:1033-1046 (instance level padding)", "mles": [[1, 0], [3, 0], [5, 0]], "num_instances": 1,
         "num_limbs": 2, "default": 1, "expected": [[1, 3, 5, 1], [1, 1, 1, 1]]},
        {"ref": "utils.rs:1049-1065 (edge: one instance)", "mles": [[2], [3]], "num_instances": 1, "num_limbs": 2,
         "default": 1, "expected": [[2, 3], [1, 1]]},
    ],
    "infer_tower_logup_witness": [
        {  # utils.rs:1067-1194 ; p = None
            "ref": "ceno_zkvm/src/scheme/utils.rs:1067-1194",
            "p": None,
            "q": [[1, 2, 3, 4], [5, 6, 7, 8]],
            "num_layers": 3,
            # layers listed output -> input; each layer [p1, p2, q1, q2]
            "layers": [
                [[(1 + 5) * (3 * 7) + (3 + 7) * 5], [(2 + 6) * (4 * 8) + (4 + 8) * (2 * 6)], [(3 * 7) * 5],
                 [(4 * 8) * (2 * 6)]],
                [[1 + 5, 2 + 6], [3 + 7, 4 + 8], [5, 2 * 6], [3 * 7, 4 * 8]],
                [[1, 1, 1, 1], [1, 1, 1, 1], [1, 2, 3, 4], [5, 6, 7, 8]],
            ],
        }
    ],
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tower_witness.json"), "w") as f:
    json.dump(golden, f, indent=1)
print("ok")
