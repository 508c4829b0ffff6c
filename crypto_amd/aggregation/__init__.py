"""SnarkPack proof aggregation — mirror of /root/reference/legogroth16/src/aggregation/ over the C ABI.

Host logic (transcript, polynomial bookkeeping, protocol flow) follows the reference module for module; every group
operation runs on the GPU entry points: multi_pairing (dgpu_multi_miller_loop + dgpu_final_exponentiation), G1/G2 MSM
(dgpu_msm_*), the GIPA folding step (dgpu_g1/g2_mul_add_batch), SRS powers (dgpu_window_table_*), batched pairing
checks (crypto_amd.pairing_check.RandomizedPairingChecker).
"""
from .transcript import MerlinTranscript, NativeMerlinTranscript          # noqa: F401
from .srs import GenericSRS, setup_fake_srs, Key, PairCommitment   # noqa: F401
from .groth16 import aggregate_proofs, verify_aggregate_proof, AggregationError   # noqa: F401
