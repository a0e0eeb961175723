"""``detzero_utils`` as the reference's detection tools import it (utils/detzero_utils/): re-exports of detzero_amd."""
