"""``detzero_det`` as the reference's detection tools import it (detection/detzero_det/): re-exports of detzero_amd."""
