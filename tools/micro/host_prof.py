"""Where the HOST time of one configs[2] step goes (cProfile over 5 eager steps; the step is
~3,100 launches and on some boxes the host, not the GPU, sets its pace)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
import bench
from naive_gpt import utils

dev = torch.device('cuda:0')
model = bench.build_model('sparse', dev, bench.LAYERS)
tuner = utils.SparseTuner(model)
gen = torch.Generator(device=dev).manual_seed(1)
batch = torch.randint(3, bench.VOCAB, [16, bench.S + 2], device=dev, generator=gen)
for _ in range(5):
    tuner.training_step(batch)
torch.cuda.synchronize()
# host pace alone: enqueue time of a step (no synchronisation inside)
for _ in range(3):
    t0 = time.perf_counter(); tuner.training_step(batch); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('enqueue %.1f ms, then drained after %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
# forward / backward / update split of the enqueue time
for _ in range(2):
    t0 = time.perf_counter()
    tuner.arm_triggers()
    loss = tuner.step_loss(batch[:, 1:-1], target=batch[:, 2:]) + tuner.aux_weight * tuner.aux_loss()
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    tuner.apply_gradients()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    print('forward %.1f ms, backward %.1f ms, update %.1f ms (enqueue)' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):      # backward functions in THIS thread: profiled
    pr.enable()
    for _ in range(5):
        tuner.training_step(batch)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats('tottime').print_stats(60)
st.sort_stats('cumtime').print_stats(70)
