"""HTTP front of the MI355X build: the reference's REST surface (api.py:35-300) over this package's IndexTTS.

    python index-tts-lora_amd/api.py --model_dir checkpoints --config checkpoints/config.yaml [--port 7859]

Same routes, request fields, defaults and status codes as the reference server, so a client written against it keeps
working:
    GET  /models          {"models": [{name, filename, type}], "current_model": ...}          (api.py:97-116)
    POST /model/reload    {"model_filename": ...} -> hot-swaps tts.gpt (404 when the file is missing)  (api.py:118-175)
    POST /tts             text + prompt_audio (upload) | prompt_audio_path, infer_mode fast|normal, speaker_id, seed and the
                          generation knobs -> audio/wav, header X-Seed; 400 without a prompt, 404 for a missing prompt path,
                          503 before the engine is up, 500 with the error text otherwise                (api.py:177-300)
/tts accepts the reference's multipart form (parsed with the standard library: python-multipart is optional), a urlencoded
form, or a JSON body with the same field names (the reference's TTSRequest model, api.py:35-50).  One engine instance per
process, requests are serialised by a lock (the reference's endpoints are effectively serial as well: one global
IndexTTS with per-instance caches)."""
from __future__ import annotations

import argparse
import json
import os
import random
import tempfile
import threading
import time
import urllib.parse
from email.parser import BytesParser
from typing import Optional

from fastapi import FastAPI, HTTPException, Request
from fastapi.responses import Response
from pydantic import BaseModel


class TTSRequest(BaseModel):
    """Field names and defaults of the reference's /tts form (api.py:178-195; the JSON model api.py:35-50 has the same names)."""
    text: str
    prompt_audio_path: Optional[str] = None
    infer_mode: str = "fast"   # 'normal' | 'fast'
    speaker_id: Optional[str] = None
    seed: Optional[int] = None
    max_text_tokens_per_sentence: int = 120
    sentences_bucket_max_size: int = 4
    do_sample: bool = True
    top_p: float = 0.8
    top_k: int = 30
    temperature: float = 0.3
    repetition_penalty: float = 10.0
    length_penalty: float = 0.0
    max_mel_tokens: int = 600


class ModelReloadRequest(BaseModel):
    model_filename: str


def _parse_multipart(body: bytes, content_type: str):
    """multipart/form-data with the standard library: {name: str | (filename, bytes)}."""
    msg = BytesParser().parsebytes(b"Content-Type: " + content_type.encode() + b"\r\nMIME-Version: 1.0\r\n\r\n" + body)
    out = {}
    for part in msg.get_payload() if msg.is_multipart() else []:
        name = part.get_param("name", header="content-disposition")
        if name is None:
            continue
        filename = part.get_param("filename", header="content-disposition")
        data = part.get_payload(decode=True) or b""
        out[name] = (filename, data) if filename is not None else data.decode("utf-8")
    return out


def _inside(path: str, roots) -> bool:
    """True when `path` (symlinks resolved) lies in one of the directories `roots`."""
    real = os.path.realpath(path)
    for r in roots:
        rr = os.path.realpath(r)
        if real == rr or real.startswith(rr.rstrip(os.sep) + os.sep):
            return True
    return False


def create_app(tts=None, model_dir="checkpoints", config_path="checkpoints/config.yaml", device=None, use_fp16=True,
               finetune_dir=os.path.join("finetune_models", "checkpoints"), output_dir=os.path.join("outputs", "api"),
               prompt_dir="prompts"):
    """`tts`: a ready IndexTTS (tests, embedding) or None to build one from model_dir / config_path on first use.
    Server-side paths a client may name are confined (the reference opens whatever it is given, api.py:125-134, :218-226):
    /model/reload only loads files inside model_dir / finetune_dir, /tts prompt_audio_path only reads files inside
    prompt_dir / model_dir; anything else answers 403."""
    from indextts.infer import IndexTTS, set_seed

    app = FastAPI(title="IndexTTS API (MI355X build)", version="1.0.0")
    state = {"tts": tts}
    lock = threading.Lock()
    model_roots = [model_dir, finetune_dir]
    prompt_roots = [prompt_dir, model_dir]

    def engine():
        if state["tts"] is None:
            if not os.path.exists(model_dir):
                raise HTTPException(status_code=503, detail=f"model directory {model_dir} does not exist")
            state["tts"] = IndexTTS(model_dir=model_dir, cfg_path=config_path, device=device, is_fp16=use_fp16)
        return state["tts"]

    app.state.engine = engine

    @app.get("/models")
    def list_models():
        models = []
        if os.path.exists(os.path.join(model_dir, "gpt.pth")):
            models.append({"name": "Default (gpt.pth)", "filename": "gpt.pth", "type": "base"})
        if os.path.exists(finetune_dir):
            for f in sorted(os.listdir(finetune_dir)):
                if f.endswith(".pth"):
                    models.append({"name": f"Finetuned - {f}", "filename": os.path.join(finetune_dir, f), "type": "finetune"})
        t = state["tts"]
        return {"models": models, "current_model": os.path.basename(t.gpt_path) if t is not None and t.gpt_path else "None"}

    @app.post("/model/reload")
    def reload_model(request: ModelReloadRequest):
        model_path = request.model_filename
        if not os.path.isabs(model_path):
            if os.path.exists(os.path.join(model_dir, model_path)):
                model_path = os.path.join(model_dir, model_path)
            elif not os.path.exists(model_path):
                raise HTTPException(status_code=404, detail=f"model file {model_path} does not exist")
        elif not os.path.exists(model_path):
            raise HTTPException(status_code=404, detail=f"model file {model_path} does not exist")
        if not _inside(model_path, model_roots):
            raise HTTPException(status_code=403, detail="model files are only loaded from the model / fine-tune directories")
        try:
            with lock:
                engine().reload_gpt(model_path)
            return {"status": "success", "message": f"switched to model: {os.path.basename(model_path)}"}
        except HTTPException:
            raise
        except Exception as e:  # noqa: BLE001
            raise HTTPException(status_code=500, detail=str(e))

    @app.post("/tts")
    async def text_to_speech(request: Request):
        ctype = request.headers.get("content-type", "")
        body = await request.body()
        upload = None
        try:
            if ctype.startswith("application/json"):
                fields = json.loads(body or b"{}")
            elif ctype.startswith("multipart/form-data"):
                fields = _parse_multipart(body, ctype)
                up = fields.pop("prompt_audio", None)
                if isinstance(up, tuple) and up[1]:
                    upload = up
            else:
                fields = {k: v[-1] for k, v in urllib.parse.parse_qs(body.decode("utf-8")).items()}
            fields = {k: (None if isinstance(v, str) and v == "" and k in ("seed", "speaker_id", "prompt_audio_path") else v)
                      for k, v in fields.items()}
            req = TTSRequest(**fields)
        except HTTPException:
            raise
        except Exception as e:  # noqa: BLE001  (validation: FastAPI answers 422 for a malformed form)
            raise HTTPException(status_code=422, detail=str(e))
        if state["tts"] is None and not os.path.exists(model_dir):
            raise HTTPException(status_code=503, detail="the TTS engine is not initialised")
        if upload is None and not req.prompt_audio_path:
            raise HTTPException(status_code=400, detail="a reference audio is required (upload a file or give a path)")
        actual_seed = random.randint(0, 2 ** 32 - 1) if req.seed is None or req.seed == -1 else int(req.seed)
        tmp_path = None
        try:
            if upload is not None:
                suffix = os.path.splitext(upload[0] or "")[1] or ".wav"
                with tempfile.NamedTemporaryFile(delete=False, suffix=suffix) as tmp:
                    tmp.write(upload[1])
                    tmp_path = prompt = tmp.name
            else:
                if not os.path.exists(req.prompt_audio_path):
                    raise HTTPException(status_code=404, detail=f"reference audio {req.prompt_audio_path} does not exist")
                if not _inside(req.prompt_audio_path, prompt_roots):
                    raise HTTPException(status_code=403, detail="server-side prompts are only read from the prompt directory")
                prompt = req.prompt_audio_path
            os.makedirs(output_dir, exist_ok=True)
            name = f"gen_{int(time.time())}_{os.urandom(2).hex()}.wav"
            out_path = os.path.join(output_dir, name)
            kwargs = dict(do_sample=req.do_sample, top_p=req.top_p, top_k=req.top_k if req.top_k > 0 else 30,
                          temperature=req.temperature, repetition_penalty=req.repetition_penalty,
                          length_penalty=req.length_penalty, num_beams=3, max_mel_tokens=req.max_mel_tokens)
            with lock:
                t = engine()
                set_seed(actual_seed)
                if req.infer_mode == "fast":
                    t.infer_fast(audio_prompt=prompt, text=req.text, output_path=out_path,
                                 max_text_tokens_per_sentence=req.max_text_tokens_per_sentence,
                                 sentences_bucket_max_size=req.sentences_bucket_max_size, **kwargs)
                else:
                    t.infer(audio_prompt=prompt, text=req.text, output_path=out_path,
                            max_text_tokens_per_sentence=req.max_text_tokens_per_sentence, speaker_id=req.speaker_id, **kwargs)
            if not os.path.exists(out_path):
                raise RuntimeError("no audio was produced")
            with open(out_path, "rb") as f:
                data = f.read()
            return Response(content=data, media_type="audio/wav",
                            headers={"Content-Disposition": f"attachment; filename={name}", "X-Seed": str(actual_seed)})
        except HTTPException:
            raise
        except Exception as e:  # noqa: BLE001
            raise HTTPException(status_code=500, detail=str(e))
        finally:
            if tmp_path and os.path.exists(tmp_path):
                os.remove(tmp_path)

    return app


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="IndexTTS API server (MI355X build)")
    ap.add_argument("--host", type=str, default="127.0.0.1",
                    help="bind address (the reference binds 0.0.0.0; opt in explicitly: this server loads checkpoints)")
    ap.add_argument("--port", type=int, default=7859)
    ap.add_argument("--model_dir", type=str, default="checkpoints")
    ap.add_argument("--config", type=str, default="checkpoints/config.yaml")
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--no-fp16", action="store_true")
    ap.add_argument("--prompt_dir", type=str, default="prompts", help="directory /tts prompt_audio_path may read from")
    a = ap.parse_args()
    import uvicorn
    uvicorn.run(create_app(None, a.model_dir, a.config, a.device, not a.no_fp16, prompt_dir=a.prompt_dir), host=a.host, port=a.port)
