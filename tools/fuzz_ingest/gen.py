import sys, os, wave, struct
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from flac_writer import encode_flac
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)   # usage: gen.py N [seed]
os.makedirs("cases", exist_ok=True)
def speech(n, seed):
    r = np.random.default_rng(seed)
    t = np.arange(n)
    return (3000*np.sin(t*0.05) + 800*r.standard_normal(n)).astype(np.int64)
seeds = []
seeds.append(bytes(encode_flac(speech(9000, 1), kinds=("fixed2","lpc"), blocksize=1152, porder=2)))
seeds.append(bytes(encode_flac(np.stack([speech(5000,2), speech(5000,3)],1), kinds=("fixed2","lpc"), stereo="mid_side", blocksize=1152)))
seeds.append(bytes(encode_flac(speech(6000, 4), bps=16, kinds=("fixed1","constant","verbatim"), blocksize=2048, wasted=True)))
# wav seed
import io
b = io.BytesIO()
w = wave.open(b, "wb"); w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(speech(4000,5).astype("<i2").tobytes()); w.close()
seeds.append(b.getvalue())
n = 0
for si, s in enumerate(seeds):
    for k in range(int(sys.argv[1])):
        blob = bytearray(s)
        mode = k % 4
        if mode == 0:   # flip a few random bytes anywhere
            for _ in range(rng.integers(1, 6)):
                blob[rng.integers(0, len(blob))] = rng.integers(0, 256)
        elif mode == 1: # corrupt header region
            for _ in range(rng.integers(1, 4)):
                blob[rng.integers(0, min(64, len(blob)))] = rng.integers(0, 256)
        elif mode == 2: # truncate
            blob = blob[: rng.integers(0, len(blob))]
        else:           # bit flips in the frame area
            for _ in range(rng.integers(1, 10)):
                i = rng.integers(min(42, len(blob)-1), len(blob)); blob[i] ^= 1 << rng.integers(0, 8)
        ext = "wav" if si == 3 else "flac"
        open(f"cases/c{si}_{k}.{ext}", "wb").write(bytes(blob)); n += 1
print(n)
