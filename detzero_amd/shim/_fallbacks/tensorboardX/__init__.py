"""Fall-back for tensorboardX (absent from the ROCm image): a SummaryWriter that appends scalars to
``<log_dir>/scalars.jsonl`` and ignores images.  Only found when the real package is not installed
(``detzero_amd.shim.install`` appends this directory to the END of sys.path)."""
import json
import os


class SummaryWriter:
    def __init__(self, log_dir=None, **kwargs):
        self.log_dir = log_dir
        self._f = None
        if log_dir is not None:
            os.makedirs(log_dir, exist_ok=True)
            self._f = open(os.path.join(log_dir, 'scalars.jsonl'), 'a')

    def add_scalar(self, tag, value, global_step=None, **kwargs):
        if self._f is not None:
            self._f.write(json.dumps({'tag': tag, 'value': float(value), 'step': global_step if global_step is None else float(global_step)}) + '\n')
            self._f.flush()

    def add_image(self, *args, **kwargs):
        pass

    def flush(self):
        if self._f is not None:
            self._f.flush()

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None
