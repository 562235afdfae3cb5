import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# Tests that put several handles of a node-sharded group into ONE process (one thread per rank on the box's single GPU) run kernels that wait for each other — the victim
# actions' kernels stay resident while the ranks exchange a wave's outcomes — so every stream needs a hardware queue of its own: the runtime's default of four queues per
# process would put two such streams behind each other.  (One process per GPU, as the product runs, holds two streams.)  Read at HIP initialisation, hence here.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
