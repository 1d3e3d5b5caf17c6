"""Which of torch's own GPU kernels (libtorch_hip.so of this image, gfx950 code objects) contain packed f32 VALU
instructions whose LOW result takes the HIGH half of a source (op_sel / op_sel_hi forms of v_pk_mul_f32 / v_pk_fma_f32 /
v_pk_add_f32)?  Those are the forms that returned wrong results beside v_mfma_f32_16x16x32_bf16 on MI355X
(profiles/README.md, "A packed multiply beside v_mfma_f32_16x16x32_bf16"); torch's kernels run beside this library's
bf16 convolutions whenever a side stream is busy (dropout, fills, Adam).  CPU only: .hip_fatbin -> compressed offload
bundles (CCOB) -> clang-offload-bundler -> llvm-objdump -d, one code object at a time.

  python scripts/study/scan_torch_packed_f32.py [kernel-name substrings ...] > profiles/r06_torch_packed_f32_scan.txt
"""
import os, re, subprocess, sys, tempfile, collections

LLVM = '/opt/rocm/lib/llvm/bin'
import torch
lib = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libtorch_hip.so')
want = sys.argv[1:]
tmp = tempfile.mkdtemp(prefix='torchscan_')
fat = os.path.join(tmp, 'fatbin.bin')
subprocess.run(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', lib, fat], check=True)
data = open(fat, 'rb').read()
starts = [m.start() for m in re.finditer(b'CCOB', data) if data[m.start() + 24:m.start() + 28] == b'\x28\xb5\x2f\xfd']
print('%s: %d bytes of .hip_fatbin, %d compressed offload bundles' % (lib, len(data), len(starts)))
pk = re.compile(r'\b(v_pk_(?:mul|fma|add)_f32)\b(.*)')
per_kernel = collections.defaultdict(lambda: collections.Counter())
tot = collections.Counter()
n_obj = 0
for i, a in enumerate(starts):
    import struct
    _ver, _method, total, _unc = struct.unpack_from('<HHII', data, a + 4)      # CCOB v2 header: the bundle's exact size
    piece = os.path.join(tmp, 'b.bin')
    open(piece, 'wb').write(data[a:a + total])
    co = os.path.join(tmp, 'b.co')
    r = subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + piece,
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co, '--allow-missing-bundles'],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
        continue
    n_obj += 1
    p = subprocess.Popen([LLVM + '/llvm-objdump', '-d', '--no-show-raw-insn', '-C', co], stdout=subprocess.PIPE, text=True, errors='replace')
    sym = '?'
    for line in p.stdout:
        if line.endswith('>:\n'):
            sym = line[line.find('<') + 1:-3]
            continue
        m = pk.search(line)
        if m:
            form = m.group(1) + (' op_sel' if 'op_sel:' in m.group(2) else '') + (' op_sel_hi' if 'op_sel_hi:' in m.group(2) else '')
            tot[form] += 1
            if 'op_sel' in form:
                per_kernel[sym][form] += 1
    p.wait()
    os.remove(co)
print('%d gfx950 code objects disassembled' % n_obj)
print('packed f32 instructions by form:')
for k, v in sorted(tot.items()):
    print('  %-40s %d' % (k, v))
print('kernels with an op_sel / op_sel_hi form: %d' % len(per_kernel))
if os.environ.get('SCAN_FULL'):         # every such kernel with its forms, one line each, for grep
    with open(os.environ['SCAN_FULL'], 'w') as fh:
        for sym, c in sorted(per_kernel.items()):
            fh.write('%6d  %s   %s\n' % (sum(c.values()), sym, dict(c)))
shown = 0
for sym, c in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1].values())):
    hit = (not want) or any(w in sym for w in want)
    if hit and shown < 400:
        print('  %6d  %s   %s' % (sum(c.values()), sym[:220], dict(c)))
        shown += 1
for w in want:
    n = sum(1 for s in per_kernel if w in s)
    print('kernels matching %r with such a form: %d' % (w, n))
