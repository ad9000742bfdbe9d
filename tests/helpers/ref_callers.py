"""TEST INFRASTRUCTURE (build container only: needs /root/reference).  Run as a subprocess by tests/test_ref_interop_cpu.py.

The drop-in boundary seen from the reference's OWN callers:
  1. `utils.write_loss` (utils.py:277-305) reflects over the trainer: the reference's function is run, unmodified, over a
     reference `Council_Trainer` and over ours (both freshly constructed from the same config) with a recording writer --
     the sets of tags and the value types must be equal;
  2. every `trainer.<attr>` use in train.py (:87,103,241-250,289-297,350-351,380,388-399) and test_on_folder.py (:67-118) is
     collected by walking the AST: the attribute must exist on our trainer and every call must bind against our signature
     with the same positional / keyword pattern;
  3. the member-list uses of test_on_folder.py (`gen_a2b_s[i].encode / .decode / .load_state_dict / .cuda_device`)."""
import ast
import copy
import inspect
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402
import yaml  # noqa: E402

REF = ref_shim.REFERENCE_ROOT


class Recorder:
    def __init__(self):
        self.tags = {}

    def add_scalar(self, tag, value, it):
        self.tags[tag] = ('scalar', type(value).__name__)

    def add_scalars(self, tag, values, it):
        self.tags[tag] = ('scalars', tuple(sorted((k, type(v).__name__) for k, v in values.items())))


def trainer_uses(path):
    """[(attr, call node or None)] for every `trainer.<attr>` in the file; subscripted member lists give (attr, 'member', name)."""
    tree = ast.parse(open(path).read())
    uses, members = [], []
    parents = {}
    for node in ast.walk(tree):
        for ch in ast.iter_child_nodes(node):
            parents[ch] = node
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == 'trainer':
            par = parents.get(node)
            call = par if isinstance(par, ast.Call) and par.func is node else None
            uses.append((node.attr, call, node.lineno))
            # trainer.gen_a2b_s[i].encode ...
            if isinstance(par, ast.Subscript):
                pp = parents.get(par)
                if isinstance(pp, ast.Attribute):
                    members.append((node.attr, pp.attr, node.lineno))
    return uses, members


def main():
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female_council_folder.yaml")))
    cfg['gen'].update(dim=4, mlp_dim=8, n_res=1)
    cfg['dis'].update(dim=4)
    cfg['council']['council_size'] = 2
    cfg['iteration'] = 0
    import council_gan_amd as cga
    import utils as RU                      # the reference's utils.py (write_loss)
    Trainer = ref_shim.reference_trainer_cls()
    torch.manual_seed(1)
    ref = Trainer(copy.deepcopy(cfg), 'cpu')
    cga.seed_everything(1)
    ours = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')      # host-side construction only

    # 1. write_loss
    ra, rb = Recorder(), Recorder()
    RU.write_loss(0, ref, ra)
    RU.write_loss(0, ours, rb)
    only_ref = sorted(set(ra.tags) - set(rb.tags))
    only_ours = sorted(set(rb.tags) - set(ra.tags))
    assert not only_ref, "logged by the reference trainer, missing on ours: %s" % only_ref
    assert not only_ours, "logged by ours only: %s" % only_ours
    for k in ra.tags:
        assert ra.tags[k] == rb.tags[k], (k, ra.tags[k], rb.tags[k])
    n_tags = len(ra.tags)

    # 2. trainer.<attr> uses in the reference's drivers
    n_uses = 0
    for fn in ("train.py", "test_on_folder.py"):
        uses, members = trainer_uses(os.path.join(REF, fn))
        assert uses, fn
        for attr, call, line in uses:
            assert hasattr(ours, attr), "%s:%d uses trainer.%s" % (fn, line, attr)
            assert hasattr(ref, attr), (fn, line, attr)
            if call is not None:
                sig = inspect.signature(getattr(ours, attr))
                args = [None] * len(call.args)
                kwargs = {k.arg: None for k in call.keywords if k.arg}
                try:
                    sig.bind(*args, **kwargs)
                except TypeError as e:
                    raise AssertionError("%s:%d trainer.%s(%d positional, %s) does not bind against ours %s: %s"
                                         % (fn, line, attr, len(args), sorted(kwargs), sig, e))
                inspect.signature(getattr(ref, attr)).bind(*args, **kwargs)
            n_uses += 1
        # 3. member-list uses
        for lst, mattr, line in members:
            for m_ours, m_ref in zip(getattr(ours, lst), getattr(ref, lst)):
                assert hasattr(m_ours, mattr), "%s:%d uses trainer.%s[i].%s" % (fn, line, lst, mattr)
                assert callable(getattr(m_ours, mattr)) == callable(getattr(m_ref, mattr)), (lst, mattr)
            n_uses += 1
    print("CALLERS_OK write_loss tags=%d trainer uses=%d" % (n_tags, n_uses))


if __name__ == "__main__":
    main()
