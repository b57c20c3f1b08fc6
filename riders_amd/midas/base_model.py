"""Same load/save contract as the reference's modules/midas/base_model.py:5-34 (bare state_dict; also accepts
{'optimizer','model'} checkpoints and lightning 'state_dict' with a 'model.' prefix)."""
import torch


class BaseModel(torch.nn.Module):
    def load(self, path):
        parameters = torch.load(path, map_location=torch.device('cpu'))
        if "optimizer" in parameters:
            parameters = parameters["model"]
        if 'state_dict' in parameters:
            state_dict = parameters['state_dict']
            new_state_dict = {}
            for k, v in state_dict.items():
                new_key = k[len("model."):] if k.startswith("model.") else k
                new_state_dict[new_key] = v
            parameters = new_state_dict
        self.load_state_dict(parameters)

    def save(self, path):
        torch.save(self.state_dict(), path)
