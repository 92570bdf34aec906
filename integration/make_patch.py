#!/usr/bin/env python3
"""Regenerates integration/physher-device.patch: the maintainer's form of seam A -- physher's treelikelihood.[ch]
gain a table of accelerator hooks, and each of the eleven functions integration/physher_device.c otherwise interposes by symbol
hands its arguments to the table once (and runs its own body when the hook calls it back).  With the patch applied the binding is
built with -DPHYSHER_DEVICE_PATCHED and registers its functions in that table at load time: no LD_PRELOAD, no link-order
dependence, works against a static libphyc, -Bsymbolic, -fno-semantic-interposition or LTO builds.

usage: make_patch.py PHYSHER_SRC > physher-device.patch     (reads PHYSHER_SRC/src/phyc/treelikelihood.{h,c}; writes nothing there)
"""
import difflib
import os
import re
import sys

HOOKS = [  # (field, return type, reference function, argument names)
    ("new_model", "Model *", "new_TreeLikelihoodModel", "name, tlk, tree, m, sm, bm"),
    ("new_model_from_json", "Model *", "new_TreeLikelihoodModel_from_json", "node, hash"),
    ("allocate_storage", "void", "allocate_storage", "tlk, index"),
    ("free_internals", "void", "free_SingleTreeLikelihood_internals", "tlk"),
    ("update_uppers", "void", "SingleTreeLikelihood_update_uppers", "tlk"),
    ("update_upper_partials", "void", "update_upper_partials", "tlk, node, include_root_freqs"),
    ("calculate_dlnl_dQ", "double", "calculate_dlnl_dQ", "tlk, index, pattern_likelihoods"),
    ("gradient_cat_branch_lengths", "void", "gradient_cat_branch_lengths", "tlk, branch_grandient, pattern_likelihoods"),
    ("gradient_pinv_sitemodel", "void", "gradient_pinv_sitemodel", "tlk, branch_gradient, branch_lengths, gradient"),
    ("gradient_pinv_W_sitemodel", "void", "gradient_pinv_W_sitemodel", "tlk, branch_gradient, branch_lengths, gradient"),
    ("calculate_gradient", "void", "TreeLikelihood_calculate_gradient", "model, grads"),
]

HEADER_BLOCK = '''
/* Accelerator hooks (physher_device.c).  A device backend installs a table; every hooked function hands its arguments to the
 * table once and runs its own body when the hook calls it back (for objects that are not on a device, or around the body). */
typedef struct TreeLikelihoodDeviceHooks {
	Model *(*new_model)(const char *, SingleTreeLikelihood *, Model *, Model *, Model *, Model *);
	Model *(*new_model_from_json)(json_node *, Hashtable *);
	void (*allocate_storage)(SingleTreeLikelihood *, size_t);
	void (*free_internals)(SingleTreeLikelihood *);
	void (*update_uppers)(SingleTreeLikelihood *);
	void (*update_upper_partials)(SingleTreeLikelihood *, Node *, bool);
	double (*calculate_dlnl_dQ)(SingleTreeLikelihood *, int, const double *);
	void (*gradient_cat_branch_lengths)(SingleTreeLikelihood *, double *, const double *);
	void (*gradient_pinv_sitemodel)(SingleTreeLikelihood *, const double *, const double *, double *);
	void (*gradient_pinv_W_sitemodel)(SingleTreeLikelihood *, const double *, const double *, double *);
	void (*calculate_gradient)(Model *, double *);
} TreeLikelihoodDeviceHooks;
extern const TreeLikelihoodDeviceHooks *treelikelihood_device_hooks;
'''

SOURCE_BLOCK = '''
const TreeLikelihoodDeviceHooks *treelikelihood_device_hooks = NULL;
/* one hand-over per call: the hook may call the function back, which then runs its own body */
#define DEVICE_HOOK(field, call) do { static __thread int depth_; \\
	if (treelikelihood_device_hooks && treelikelihood_device_hooks->field && !depth_) { depth_++; call; depth_--; return; } } while (0)
#define DEVICE_HOOK_VALUE(field, type, call) do { static __thread int depth_; \\
	if (treelikelihood_device_hooks && treelikelihood_device_hooks->field && !depth_) { depth_++; type r_ = call; depth_--; return r_; } } while (0)
'''


def patched(src_dir):
    h = open(os.path.join(src_dir, "treelikelihood.h")).read()
    c = open(os.path.join(src_dir, "treelikelihood.c")).read()
    # header: the table goes behind the last declaration the binding hooks (before the closing include guard)
    k = h.rindex("#endif")
    h2 = h[:k] + HEADER_BLOCK.lstrip("\n") + "\n" + h[k:]
    # source: the table's definition in front of the file's first declaration, a hand-over as the first statement of each function
    first = re.search(r"^void TreeLikelihood_calculate_gradient\(.*\);[ \t]*\n", c, re.M)
    if not first:
        raise SystemExit("anchor declaration not found")
    c2 = c[:first.start()] + SOURCE_BLOCK.lstrip("\n") + "\n" + c[first.start():]
    for field, ret, fn, args in HOOKS:
        m = re.search(r"^(?:Model\s*\*|void|double)\s*" + re.escape(fn) + r"\s*\([^)]*\)\s*\{[ \t]*\n", c2, re.M)
        if not m:
            raise SystemExit(f"definition of {fn} not found")
        call = f"treelikelihood_device_hooks->{field}({args})"
        line = f"\tDEVICE_HOOK({field}, {call});\n" if ret == "void" else f"\tDEVICE_HOOK_VALUE({field}, {ret.strip()}, {call});\n"
        c2 = c2[:m.end()] + line + c2[m.end():]
    return (h, h2), (c, c2)


def main():
    src = os.path.join(sys.argv[1], "src", "phyc")
    (h, h2), (c, c2) = patched(src)
    out = []
    for name, a, b in (("treelikelihood.h", h, h2), ("treelikelihood.c", c, c2)):
        out += difflib.unified_diff(a.splitlines(True), b.splitlines(True), f"a/src/phyc/{name}", f"b/src/phyc/{name}", n=1)
    sys.stdout.write("".join(out))


if __name__ == "__main__":
    main()
