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
                if k.startswith("model."):      # modules/midas/base_model.py:17-21: keys WITHOUT the prefix are dropped, not kept
                    new_state_dict[k[len("model."):]] = v
            parameters = new_state_dict
        self.load_state_dict(parameters)
        from .. import engine
        engine.refresh_packed()   # cached MFMA operands follow the loaded weights in place

    def save(self, path):
        torch.save(self.state_dict(), path)
