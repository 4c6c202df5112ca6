"""Attack base class: same public surface as the reference's PointNet/attacks/torchattacks/attack.py
(:4-195) -- name/model bookkeeping, eval/train switching around the call, attack-mode and
return-type setters -- written for the libpsg-backed models."""
import torch

_MODES = ("default", "targeted", "least_likely")


class Attack(object):
    def __init__(self, name, model):
        self.attack = name
        self.model = model
        self.model_name = str(model).split("(")[0]
        self.training = model.training
        self.device = next(model.parameters()).device
        self._targeted = 1
        self._attack_mode = "default"
        self._return_type = "float"
        self._target_map_function = lambda images, labels: labels

    def forward(self, *input):
        raise NotImplementedError

    # ---- mode plumbing (attack.py:24-57 of the reference)
    def set_attack_mode(self, mode, target_map_function=None):
        if self._attack_mode == "only_default":
            raise ValueError("Changing attack mode is not supported in this attack method.")
        if mode == "targeted" and target_map_function is None:
            raise ValueError("Please give a target_map_function, e.g., lambda images, labels:(labels+1)%10.")
        if mode not in _MODES:
            raise ValueError(mode + " is not a valid mode. [Options : default, targeted, least_likely]")
        self._attack_mode = mode
        self._targeted = 1 if mode == "default" else -1
        if mode == "default":
            self._transform_label = self._get_label
        elif mode == "targeted":
            self._target_map_function = target_map_function
            self._transform_label = self._get_target_label
        else:
            self._transform_label = self._get_least_likely_label

    def set_return_type(self, type):
        if type not in ("float", "int"):
            raise ValueError(type + " is not a valid type. [Options : float, int]")
        self._return_type = type

    def _transform_label(self, images, labels):
        return labels

    def _get_label(self, images, labels):
        return labels

    def _get_target_label(self, images, labels):
        return self._target_map_function(images, labels)

    def _get_least_likely_label(self, images, labels):
        outputs = self.model(images)
        outputs = outputs[0] if isinstance(outputs, tuple) else outputs
        return torch.min(outputs.data, 1)[1].detach_()

    def _to_uint(self, images):
        return (images * 255).type(torch.uint8)

    def _switch_model(self):
        if self.training:
            self.model.train()
        else:
            self.model.eval()

    def save(self, data_loader, save_path=None, verbose=True):
        """Run the attack over a loader and optionally torch.save((adv, labels)) (attack.py:73-118)."""
        self.model.eval()
        xs, ys = [], []
        for images, labels in data_loader:
            xs.append(self.__call__(images, labels).cpu())
            ys.append(torch.as_tensor(labels).cpu())
        x, y = torch.cat(xs, 0), torch.cat(ys, 0)
        if save_path is not None:
            torch.save((x, y), save_path)
            if verbose:
                print("- Save Complete!")
        self._switch_model()

    def __str__(self):
        info = {k: v for k, v in self.__dict__.items() if not k.startswith("_") and k not in ("model", "attack")}
        info["attack_mode"] = "default" if self._attack_mode == "only_default" else self._attack_mode
        info["return_type"] = self._return_type
        return self.attack + "(" + ", ".join("{}={}".format(k, v) for k, v in info.items()) + ")"

    def __call__(self, *input, **kwargs):
        self.model.eval()
        images = self.forward(*input, **kwargs)
        self._switch_model()
        if self._return_type == "int":
            images = self._to_uint(images)
        return images
