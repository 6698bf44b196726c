"""Device vs host vision preprocessing on 1080p frames."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from hippomm_amd import preprocess as pp
B, H, W = 64, 1080, 1920
frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
for _ in range(3): pp.preprocess_frames_device(frames)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): pp.preprocess_frames_device(frames)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"device: {ms:.3f} ms per {B} frames of {H}x{W} = {B/ms*1e3:.0f} frames/s  ({B*H*W*3/ms/1e6:.0f} GB/s of input)")
img = frames[0].cpu().numpy()
t = time.perf_counter()
for _ in range(5):
    im = Image.fromarray(img).resize((398, 224), Image.BICUBIC).crop((87, 0, 311, 224))
    a = (np.asarray(im, dtype=np.float32).transpose(2, 0, 1) / 255.0 - 0.45) / 0.27
print(f"host PIL (resize+crop+normalise, 1 core): {(time.perf_counter()-t)/5*1e3:.2f} ms per frame")

# audio: 128 ten-second segments -> (128,3,1,128,204)
waves = [torch.randn(1, 160000) * 0.1 for _ in range(128)]
dev = torch.device("cuda")
out = pp.transform_waveforms_device(waves, dev)
clips = torch.randn(384, 32000, device="cuda") * 0.1
for _ in range(3): pp.melspec_clips_device(clips)
torch.cuda.synchronize()
e0.record()
for _ in range(10): pp.melspec_clips_device(clips)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"audio fbank: {ms:.3f} ms per 384 clips (128 segments) = {128/ms*1e3:.0f} segments/s, {384*198/ms*1e3/1e6:.2f} M frames/s")
