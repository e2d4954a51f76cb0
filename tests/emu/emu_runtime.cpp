// TEST INFRASTRUCTURE -- scheduler of the host-side SIMT emulator (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>

namespace emu {
Block* g_blk = nullptr;

asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch, .-emu_switch
)");

static const size_t STACK = 1024 * 1024;
static const size_t POISON_STACK = 96 * 1024;

static void trampoline() {
  Block* b = g_blk;
  b->body();
  Lane& l = b->lanes[b->cur];
  l.state = DONE;
  emu_switch(&l.sp, b->main_sp);
  abort();  // a finished fiber is never resumed
}

void block_here() {
  Block* b = g_blk;
  Lane& l = b->lanes[b->cur];
  l.state = BLOCKED;
  emu_switch(&l.sp, b->main_sp);
}

static bool at(const Block& b, int lane, int op) {
  return lane >= 0 && lane < (int)b.lanes.size() && b.lanes[lane].state == BLOCKED && b.lanes[lane].op == op;
}

// source lane of a DPP control word for destination lane i (-1: no source)
static int dpp_source(int ctrl, int i) {
  const int row = i & ~15, k = i & 15;
  if (ctrl >= 0x000 && ctrl <= 0x0ff) return (i & ~3) + ((ctrl >> (2 * (i & 3))) & 3);          // quad_perm
  if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl & 15; return k + n < 16 ? i + n : -1; }  // row_shl
  if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl & 15; return k >= n ? i - n : -1; }      // row_shr
  if (ctrl >= 0x121 && ctrl <= 0x12f) { const int n = ctrl & 15; return row + ((k - n) & 15); }     // row_ror
  if (ctrl == 0x130) return i < 63 ? i + 1 : -1;                                                   // wave_shl:1
  if (ctrl == 0x138) return i > 0 ? i - 1 : -1;                                                    // wave_shr:1
  if (ctrl == 0x140) return row + (15 - k);                                                        // row_mirror
  if (ctrl == 0x141) return row + (k < 8 ? 7 - k : 23 - k);                                        // row_half_mirror
  if (ctrl == 0x142) return row >= 16 ? row - 1 : -1;                                              // row_bcast:15
  if (ctrl == 0x143) return i >= 32 ? 31 : -1;                                                     // row_bcast:31
  if (ctrl >= 0x150 && ctrl <= 0x15f) return row + (ctrl & 15);                                    // row_newbcast (gfx90a+)
  fprintf(stderr, "emu: unsupported DPP control 0x%x\n", ctrl);
  abort();
}

static void resolve(Block& b) {
  const int n = (int)b.lanes.size();
  // block barrier: only completes once every live lane of the workgroup waits at one
  bool all_block_barrier = true;
  for (int i = 0; i < n; i++)
    if (b.lanes[i].state == BLOCKED && b.lanes[i].op != OP_BLOCK_BARRIER) all_block_barrier = false;
  for (int w0 = 0; w0 < n; w0 += 64) {
    const int wn = std::min(64, n - w0);
    uint64_t ballot = 0;
    for (int i = 0; i < wn; i++)
      if (at(b, w0 + i, OP_BALLOT) && b.lanes[w0 + i].val) ballot |= 1ull << i;
    {   // MFMA: snapshot the operands of the wave's lanes (inactive lanes contribute zeros)
      bool any = false;
      for (int i = 0; i < wn; i++) any = any || at(b, w0 + i, OP_MFMA);
      if (any) {
        if (b.mfma.size() < (size_t)(n + 63) / 64 * 128) b.mfma.resize((size_t)(n + 63) / 64 * 128);
        float* snap = b.mfma.data() + (size_t)(w0 / 64) * 128;
        for (int i = 0; i < 64; i++) {
          float a = 0.f, bb = 0.f;
          if (i < wn && at(b, w0 + i, OP_MFMA)) { memcpy(&a, &b.lanes[w0 + i].val, 4); memcpy(&bb, &b.lanes[w0 + i].val2, 4); }
          snap[i] = a; snap[64 + i] = bb;
        }
        for (int i = 0; i < wn; i++) if (at(b, w0 + i, OP_MFMA)) b.lanes[w0 + i].mfma = snap;
      }
    }
    for (int i = 0; i < wn; i++) {
      Lane& l = b.lanes[w0 + i];
      if (l.state != BLOCKED) continue;
      switch (l.op) {
        case OP_DPP: {
          const bool wr = ((l.row_mask >> (i >> 4)) & 1) && ((l.bank_mask >> ((i & 15) >> 2)) & 1);
          if (!wr) { l.res = l.old; break; }
          const int s = dpp_source(l.ctrl, i);
          const bool ok = s >= 0 && s < wn && at(b, w0 + s, OP_DPP) && b.lanes[w0 + s].ctrl == l.ctrl;
          l.res = ok ? b.lanes[w0 + s].val : (l.bound ? 0u : l.old);
          break;
        }
        case OP_READLANE: {
          const int s = l.sel & 63;
          l.res = at(b, w0 + s, OP_READLANE) ? b.lanes[w0 + s].val : 0xDEADBEEFu;  // an inactive lane's register: garbage
          break;
        }
        case OP_BPERMUTE: {
          const int s = l.sel & 63;
          l.res = at(b, w0 + s, OP_BPERMUTE) ? b.lanes[w0 + s].val : 0u;
          break;
        }
        case OP_SWIZZLE: {
          int s;
          if (l.ctrl & 0x8000) s = (i & ~3) + ((l.ctrl >> (2 * (i & 3))) & 3);   // quad-permute mode
          else { const int a = l.ctrl & 31, o = (l.ctrl >> 5) & 31, x = (l.ctrl >> 10) & 31; s = (i & 32) | ((((i & 31) & a) | o) ^ x); }
          l.res = at(b, w0 + s, OP_SWIZZLE) ? b.lanes[w0 + s].val : 0u;
          break;
        }
        case OP_PERMLANE16_SWAP: {
          // v_permlane16_swap vdst, src: odd rows of vdst <-> even rows of src (partner lane i ^ 16); an inactive partner
          // leaves the lane's registers untouched
          const int s = i ^ 16;
          const bool ok = at(b, w0 + s, OP_PERMLANE16_SWAP);
          l.res = l.val; l.res2 = l.val2;
          if (ok) { if (i & 16) l.res = b.lanes[w0 + s].val2; else l.res2 = b.lanes[w0 + s].val; }
          break;
        }
        case OP_PERMLANE32_SWAP: {   // upper half of vdst <-> lower half of src (partner lane i ^ 32)
          const int s = i ^ 32;
          const bool ok = at(b, w0 + s, OP_PERMLANE32_SWAP);
          l.res = l.val; l.res2 = l.val2;
          if (ok) { if (i & 32) l.res = b.lanes[w0 + s].val2; else l.res2 = b.lanes[w0 + s].val; }
          break;
        }
        case OP_BALLOT: l.res64 = ballot; break;
        default: break;
      }
    }
  }
  // group rendezvous: a lane waits until every live lane of its W-lane group has arrived
  std::vector<char> hold(n, 0);
  for (int i = 0; i < n; i++) {
    const Lane& l = b.lanes[i];
    if (l.state != BLOCKED || l.op != OP_GROUP_SYNC) continue;
    const int W = l.sel, g0 = i - (i % W);
    for (int j = g0; j < g0 + W && j < n; j++)
      if (b.lanes[j].state != DONE && !(b.lanes[j].state == BLOCKED && b.lanes[j].op == OP_GROUP_SYNC && b.lanes[j].sel == W)) hold[i] = 1;   // (a lane waiting at a
                                                                         // rendezvous of ANOTHER width -- its own 32-lane group's, inside a control step -- has not arrived at this one)
  }
  for (int i = 0; i < n; i++) {
    Lane& l = b.lanes[i];
    if (l.state != BLOCKED) continue;
    if (l.op == OP_BLOCK_BARRIER && !all_block_barrier) continue;
    if (hold[i]) continue;
    l.state = RUNNABLE;
    l.op = OP_NONE;
  }
}

void run_block(Block& b) {
  const int n = (int)b.lanes.size();
  static const bool reverse = getenv("LHW_EMU_REVERSE") && atoi(getenv("LHW_EMU_REVERSE"));
  Block* outer = g_blk;
  g_blk = &b;
  b.poisoned.clear();
  for (int i = 0; i < n; i++) {
    Lane& l = b.lanes[i];
    if (!l.stack) l.stack = (char*)malloc(STACK);
    if (poison_on()) memset(l.stack + STACK - POISON_STACK, 0xFF, POISON_STACK);   // the part of the stack a kernel's locals live in
    l.tid = dim3(i);
    l.state = RUNNABLE;
    l.op = OP_NONE;
    uintptr_t top = ((uintptr_t)(l.stack + STACK)) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8);   // fake return slot: the trampoline starts with rsp == 8 (mod 16), as after a call
    *sp = nullptr;
    *--sp = (void*)&trampoline;
    for (int k = 0; k < 6; k++) *--sp = nullptr;
    l.sp = sp;
  }
  for (;;) {
    bool ran = false, live = false;
    for (int k = 0; k < n; k++) {
      const int i = reverse ? n - 1 - k : k;
      Lane& l = b.lanes[i];
      if (l.state == RUNNABLE) {
        b.cur = i;
        emu_switch(&b.main_sp, l.sp);
        ran = true;
      }
      if (l.state != DONE) live = true;
    }
    if (!live) break;
    bool runnable = false;
    for (int i = 0; i < n; i++) if (b.lanes[i].state == RUNNABLE) runnable = true;
    if (!runnable) {
      resolve(b);
      bool any = false;
      for (int i = 0; i < n; i++) if (b.lanes[i].state == RUNNABLE) any = true;
      if (!any) { fprintf(stderr, "emu: deadlock (a lane left the kernel while others wait at __syncthreads?)\n"); abort(); }
    }
    (void)ran;
  }
  b.cur = -1;
  g_blk = outer;
}
}  // namespace emu
