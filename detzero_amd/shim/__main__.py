from . import FALLBACK_DIR, SHIM_DIR

print(SHIM_DIR + ':' + FALLBACK_DIR)
