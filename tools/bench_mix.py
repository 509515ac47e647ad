"""A config-2 batch with one region in fifty at another read length: the same work as config 2 through the RAGGED table kernel (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth
from rovaca_amd.batch import Batch

eng = capi.Engine(0)
parts = [synth.make_batch(seed=2, n_regions=980, reads=40, haps=25, read_len=150, hap_len=300),
         synth.make_batch(seed=3, n_regions=20, reads=40, haps=25, read_len=int(sys.argv[1]) if len(sys.argv) > 1 else 100, hap_len=300)]
b = Batch.concat(parts)
rb = eng.upload(b)
for _ in range(3):
    rb.sweep()
ms, ms32 = rb.sweep_timed(8)
_, _, st = rb.download()
print(f"mix pairs={b.n_pairs} launches={int(st.n_launches)} fp32 {ms32:8.3f} ms  {b.cells()/ms32/1e6:8.1f} GCUPS  (lib {os.path.basename(os.environ.get('ROVACA_AMD_HIP_LIB', 'default'))}, ug={os.environ.get('PHMM_NO_UNIFORM_GOP')})", flush=True)
rb.release()
